"""GPU: the randomised campaigns against the oracle (tools/fuzz_*.py: random grids, CV sets, boxes, particle counts, mesh sizes,
neighbour lists, host-API CV sets — DESIGN.md §3) as part of the driver-run suite: each tool runs for a few seconds with a fixed
seed in a process of its own and must finish without a deviation beyond its stated tolerances (exit code 0).  The long runs
(minutes, fresh seeds) remain developer tools."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,seconds,seed", [("fuzz_grid.py", 6, 101), ("fuzz_fused.py", 8, 102), ("fuzz_mesh.py", 10, 103),
                                               ("fuzz_ql.py", 8, 104), ("fuzz_slots.py", 6, 105), ("fuzz_host.py", 8, 106),
                                               ("fuzz_slab.py", 8, 107)])
def test_randomised_campaign(tool, seconds, seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(seconds), str(seed)], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, "%s failed:\n%s\n%s" % (tool, r.stdout[-3000:], r.stderr[-3000:])
    last = [l for l in r.stdout.splitlines() if l.startswith("fuzz_")]
    assert last and "random" in last[-1], r.stdout[-1000:]
