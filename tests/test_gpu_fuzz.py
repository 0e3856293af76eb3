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


@pytest.mark.parametrize("tool,case", [("fuzz_fused.py", "fuzz_fused_r3_case.json"), ("fuzz_host.py", "fuzz_host_r3_case.json")])
def test_round3_campaign_cases_that_changed_a_tolerance(tool, case):
    """The two cases round 3's long campaigns stopped at (profiles/r3/final_campaigns.log), as fixed regression cases: the recorded
    state of the tool's random generator reproduces exactly the failing configuration (tests/golden/fuzz_*_r3_case.json, found again
    in round 4 with FUZZ_OLD_TOL=1: fused step, case 6 519 of seed 403 — fp64 particles, accurate trigonometry, modes (0,0,0),
    (0,0,0), (-3,3,-3) at N = 40 000; host classes, case 2 188 of seed 402 — a cv.lamellar under a harmonic umbrella beside another lamellar CV
    and a mesh CV).  Each must FAIL under the tolerance formula that was in force when it was found and PASS under the one that
    replaced it, which is
      fused:  |s - s_ref| <= 1e-6 max(|s_ref|, n_wave max|a| max(1, max_modes(|h| + |k| + |l|)) / sqrt(N))   [index SUM: the fp32
              phase carries its rounding |h| + |k| + |l| times, DESIGN.md 4.1 — the old formula took the largest single index]
      host:   forces compared per unit of (grid bias factor + umbrella derivative kappa (s - cv0)), the factor the force kernel
              multiplies with (CollectiveVariable.cc:22-66), <= 2e-4 of max|F| — the old formula divided by the grid factor alone"""
    path = os.path.join(ROOT, "tests", "golden", case)
    old = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), "--replay", path], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, FUZZ_OLD_TOL="1"))
    assert old.returncode != 0 and "AssertionError" in old.stderr, "the recorded case no longer reproduces:\n" + old.stdout[-800:] + old.stderr[-800:]
    new = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), "--replay", path], capture_output=True, text=True, timeout=300)
    assert new.returncode == 0 and "replayed" in new.stdout, new.stdout[-1500:] + new.stderr[-1500:]
