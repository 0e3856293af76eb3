#!/bin/bash
# Developer tool (GPU box): the driver's call (bench.py --steps 20 --warmup 5) under runtime wait knobs, alternating fresh processes.
# usage: tools/diag/k20_env_probe.sh [rounds]     output: one line per run (official region, the ten repeats' median / min, steady state)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
R=${1:-8}
for round in $(seq 1 $R); do
  for v in "" "ROC_ACTIVE_WAIT_TIMEOUT=1000" "ROC_CPU_WAIT_FOR_SIGNAL=0"; do
    out=$(env $v timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-sub-records --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['extra']['timed_region_repeats']['ms_per_step']
print('official %.2f  repeats median %.2f min %.2f  steady %.2f' % (1e3*d['ms_per_step'], 1e3*sorted(r)[len(r)//2], 1e3*min(r), 1e3*d['extra']['steady_state']['ms_per_step']))")
    echo "[${v:-default}] $out"
  done
done
