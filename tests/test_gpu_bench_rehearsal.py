"""GPU: bench.py's N > 1 code paths rehearsed on the one-GPU box through the SAME entry the driver uses — plain
`python bench.py --gpus 2` (no launcher: bench.py starts its own ranks before it touches the GPU; with fewer GPUs than ranks it
flags the run as a rehearsal: every rank on cuda:0, control plane gloo).  Particles sharded (weak and strong scaling: the
per-step sums through the xGMI mailbox between the two processes, C++ host classes as the step loop) and multiple walkers (packed
grid increments summed over the walkers).  Numbers from such runs mean nothing; the JSON contract, the exchange chosen, zero expired
waits, the global CV values against the oracle and identical grids on all walkers do."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
        "data", "config", "roofline")


def _bench(args, **env):
    e = dict(os.environ, MTD_FUSED_STEP="0", **env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MTD_BENCH_REHEARSAL", "HSA_ENABLE_IPC_MODE_LEGACY"):
        e.pop(k, None)                      # bench.py sets what it needs itself: that is what is being tested
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(line) == 1, r.stdout[-2000:]                     # ONE line
    d = json.loads(line[0])
    for key in KEYS:
        assert key in d, key
    return d


def _sharded_checks(d, n_global):
    assert d["n_gpus"] == 2 and d["vs_baseline"] is None
    assert "rehearsal" in d["config"]
    assert d["config"]["exchange"] == "xgmi-mailbox" and d["config"]["mailbox_timeouts"] == 0, d["config"]
    assert "exchange_note" not in d["config"]
    assert d["config"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert d["config"]["particles_global"] == n_global
    assert d["value"] == pytest.approx(d["steps"] / (d["ms_per_step"] * d["steps"] * 1e-3) * n_global * 2, rel=1e-6)
    assert max(d["cv_check"]["abs_err"]) <= d["cv_check"]["tolerance"], d["cv_check"]
    assert d["self_check"]["V_rel_err"] < 1e-10 and d["self_check"]["w_rel_err"] < 1e-10
    assert "step_frac" in d["roofline"] and d["roofline"]["bound"] == "hbm"
    assert d["extra"]["steady_state"]["steps"] >= 2000


def test_bench_two_ranks_weak():
    d = _bench(["--gpus", "2", "--particles", "200000", "--steps", "40", "--warmup", "5"])
    assert d["scaling"] == "weak" and d["steps"] == 40 and d["config"]["mode"] == "sharded"
    _sharded_checks(d, 400000)


def test_bench_two_ranks_strong():
    d = _bench(["--gpus", "2", "--scaling", "strong", "--particles", "400000", "--steps", "40", "--warmup", "5"])
    assert d["scaling"] == "strong" and d["config"]["particles_per_gpu"] == 200000
    _sharded_checks(d, 400000)


def test_bench_two_walkers():
    d = _bench(["--gpus", "2", "--walkers", "--particles", "100000", "--steps", "30", "--warmup", "5"])
    assert d["scaling"] == "weak" and d["config"]["mode"] == "walkers" and d["config"]["particles_global"] == 200000
    wc = d["walker_check"]
    assert wc["grid_identical_on_all_ranks"] is True and wc["grid_sum"] > 0.0, wc
    # every deposit of a walker step adds the increments of BOTH walkers: more than one hill's worth of bias per counted hill
    assert wc["hills_rank0"] >= 30
    assert d["value"] == pytest.approx(30 / (d["ms_per_step"] * 30e-3) * 200000 * 2, rel=1e-6)


def test_bench_fallback_is_loud():
    """a mailbox that cannot be used (here: switched off) sends the ranks to the collective, and the line says so"""
    d = _bench(["--gpus", "2", "--particles", "100000", "--steps", "20", "--warmup", "5", "--no-variants"], MTD_XGMI_MAILBOX="0")
    assert d["config"]["exchange"] == "gloo" and d["config"]["exchange_note"].startswith("FALLBACK"), d["config"]
    assert max(d["cv_check"]["abs_err"]) <= d["cv_check"]["tolerance"], d["cv_check"]


def test_bench_single_gpu_through_the_same_entry():
    d = _bench(["--gpus", "1", "--steps", "20", "--warmup", "5", "--no-sub-records", "--cpu-steps", "1"])
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["config"]["mode"] == "single" and "exchange" not in d["config"]
    assert d["config"]["fast_trig"] == d["config"]["fast_trig_library_default"] == 1
    assert d["self_check"]["V_rel_err"] < 1e-10
    for k in ("steady_state", "stride100", "f64", "accurate_trig"):
        assert d["extra"][k]["ms_per_step"] > 0, k
    assert d["roofline"]["measured_peak"]["copy_512MB_GBs"] > 1000.0
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1


def test_bench_config3_two_ranks_through_the_host_classes():
    """`bench.py --config 3 --gpus 2`: cv.mesh + cv.lamellar DOMAIN-DECOMPOSED inside the C++ host classes (System::run on every rank):
    replicated mesh summed by remote loads through the mailbox's exported buffers in this rehearsal (RCCL on real GPUs), and the
    slab-decomposed mesh; the CV values the ranks agreed on against the oracle on the whole snapshot, identical state on all ranks"""
    for mesh, word in (("replicated", "xgmi-pull"), ("slab", "slab decomposition")):
        d = _bench(["--config", "3", "--gpus", "2", "--particles", "100000", "--steps", "20", "--warmup", "5", "--no-variants", "--mesh", mesh])
        assert d["n_gpus"] == 2 and d["config"]["bench_config"] == 3 and d["config"]["particles_global"] == 200000 and "rehearsal" in d["config"]
        assert d["config"]["driver"] == "host" and d["config"]["mailbox_timeouts"] == 0 and word in d["config"]["exchange"], d["config"]
        assert d["config"]["on_grid"] and d["state"]["num_gaussians"] >= 20 and d["state"]["identical_on_all_ranks"] and not d["state"]["fused"]
        assert max(d["cv_check"]["rel_err"]) <= d["cv_check"]["tolerance"], d["cv_check"]
        assert d["value"] == pytest.approx(200000 * 2 / (d["ms_per_step"] * 1e-3), rel=1e-6)
        assert d["roofline"]["bound"] == "hbm" and d["roofline"]["frac"] > 0


def test_bench_config5_two_ranks_and_one():
    """`bench.py --config 5 --gpus 2`: cv.steinhardt over z slabs with ghost layers inside the host classes; the same entry at N = 1"""
    d = _bench(["--config", "5", "--gpus", "2", "--particles", "10", "--steps", "20", "--warmup", "5", "--no-variants"])
    assert d["n_gpus"] == 2 and d["config"]["particles_global"] == 8000 and d["config"]["particles_per_gpu"] > 3000, d["config"]
    assert d["config"]["ghosts_rank0"] > 0 and d["config"]["mailbox_timeouts"] == 0 and "Q'_lm" in d["config"]["exchange"]
    assert d["config"]["on_grid"] and d["state"]["identical_on_all_ranks"] and d["state"]["num_gaussians"] >= 20
    d1 = _bench(["--config", "5", "--gpus", "1", "--particles", "10", "--steps", "20", "--warmup", "5", "--no-variants"])
    assert d1["n_gpus"] == 1 and d1["config"]["particles_global"] == 4000 and "exchange" not in d1["config"]
    assert max(d1["cv_check"]["rel_err"]) <= d1["cv_check"]["tolerance"], d1["cv_check"]
