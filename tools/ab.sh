#!/bin/bash
# Developer tool: A/B two builds of libmtd_hip.so in ONE GPU call (box-to-box variation is larger than most kernel tweaks).
# usage: tools/ab.sh <libA> <libB> [rounds]   (paths relative to the repo root; host driver uses the in-tree lib, so --driver abi)
A=$1; B=$2; R=${3:-3}
for r in $(seq 1 $R); do
  for lib in $A $B; do
    MTD_LIB_OVERRIDE=$PWD/$lib python bench.py --steps 3000 --warmup 300 --no-cpu-baseline --driver abi | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', 'us/step=%.2f' % (1e3*d['ms_per_step']), 'force_us=%.2f' % d['roofline']['avg_launch_us'])"
  done
done
