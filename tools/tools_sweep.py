#!/usr/bin/env python3
"""Developer tool: run bench.py under several kernel-geometry settings (env overrides) in one GPU call."""
import itertools, json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cases = []
for cvb, fb in itertools.product(sys.argv[1].split(","), sys.argv[2].split(",")):
    env = dict(os.environ, MTD_LAM_CV_BLOCKS=cvb, MTD_LAM_FORCE_BLOCKS=fb)
    extra = sys.argv[3:] if len(sys.argv) > 3 else []
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "600", "--warmup", "100",
                          "--no-cpu-baseline"] + extra, env=env, capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        print("cv_blocks=%s force_blocks=%s  us/step=%.2f  force_ev_us=%.2f" % (cvb, fb, 1e3 * d["ms_per_step"], d["roofline"]["avg_launch_us"]), flush=True)
    except Exception as e:
        print("cv_blocks=%s force_blocks=%s FAILED %s %s" % (cvb, fb, e, out.stderr[-300:]), flush=True)
