"""CPU: `python bench.py --gpus N` without a launcher starts its own ranks (the shape of the driver's call).  The parent must
decide that before it imports torch or touches the GPU, hand HSA_ENABLE_IPC_MODE_LEGACY=0 to the ranks (the xGMI mailbox's IPC
mapping needs it) and relay the child's exit code.  MTD_BENCH_DRY_LAUNCH=1 shows the launch instead of starting it."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dry(args, **env):
    e = dict(os.environ, MTD_BENCH_DRY_LAUNCH="1", **env)
    e.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_plain_call_launches_one_rank_per_gpu():
    d = _dry(["--gpus", "2", "--steps", "20", "--warmup", "5", "--scaling", "strong"], HIP_VISIBLE_DEVICES="0,1,2,3")
    cmd = d["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "2" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-8:] == ["--gpus", "2", "--steps", "20", "--warmup", "5", "--scaling", "strong"]
    assert os.path.basename(cmd[cmd.index("--master-port") + 2]) == "bench.py"
    assert d["torch_imported"] is False                      # the parent never loads torch (nor HIP)
    assert d["ipc"] == "0"


def test_fewer_gpus_than_ranks_is_a_flagged_rehearsal():
    d = _dry(["--gpus", "2"], HIP_VISIBLE_DEVICES="0")
    assert d["rehearsal"] == "1" and d["visible_gpus"] == 1
    # a rehearsal on one GPU is bounded (process guard of a GPU box): 8 ranks are refused, nothing is started
    e = dict(os.environ, HIP_VISIBLE_DEVICES="0")
    e.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=e, capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode == 2 and "limited to 4 ranks" in r.stderr


def test_callers_ipc_choice_is_kept():
    d = _dry(["--gpus", "2"], HIP_VISIBLE_DEVICES="0,1", HSA_ENABLE_IPC_MODE_LEGACY="1")
    assert d["ipc"] == "1"
