"""GPU: bench.py's N > 1 code path (what the driver's scaling run launches with torch.distributed.run) rehearsed with two ranks
sharing cuda:0 — MTD_BENCH_REHEARSAL=1: control plane gloo, particles sharded, the per-step sums through the xGMI mailbox between
the two processes, C++ host classes as the step loop.  Numbers from such a run mean nothing; the JSON contract, the exchange
chosen, zero expired waits and the global CV values against the oracle do."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_rehearsal():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MTD_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MTD_FUSED_STEP="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--particles", "200000", "--steps", "40",
                        "--warmup", "5"], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(line) == 1, r.stdout[-2000:]                     # rank 0 prints ONE line
    d = json.loads(line[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline"):
        assert key in d, key
    assert d["n_gpus"] == 2 and d["steps"] == 40 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["config"]["exchange"] == "xgmi-mailbox" and d["config"]["mailbox_timeouts"] == 0, d["config"]
    assert d["value"] == pytest.approx(40 / (d["ms_per_step"] * 40e-3) * 400000 * 2, rel=1e-6)
    assert max(d["cv_check"]["abs_err"]) <= d["cv_check"]["tolerance"], d["cv_check"]
    assert d["self_check"]["V_rel_err"] < 1e-10 and d["self_check"]["w_rel_err"] < 1e-10
    assert "step_frac" in d["roofline"] and d["roofline"]["bound"] == "hbm"
