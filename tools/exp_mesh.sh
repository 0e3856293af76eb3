#!/bin/bash
# Developer tool (GPU box): config 3 A/B in ONE call over builds of mesh.hip with different tile shapes / block sizes
# (EXTRA_HIPFLAGS), alternating runs; every variant build first passes tests/test_gpu_mesh.py.  Output: gpurun_out/exp_mesh.log
# usage: tools/exp_mesh.sh <steps> "<flags of variant 1>" "<flags of variant 2>" ...   ("" = the default build)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
LOG=$O/exp_mesh.log; : > $LOG
STEPS=${1:-300}; shift
run() {   # run <label> [ENV=...]
  local label=$1; shift
  local out; out=$(env "$@" timeout -k 10 200 python3 tools/bench_mesh.py $STEPS 2>&1 | grep "config 3" | tail -1)
  echo "$label: $out" | tee -a $LOG
}
for round in 1 2; do
  for V in "$@"; do
    echo "== build [$V] (round $round)" | tee -a $LOG
    touch metadynamics-plugin_amd/csrc/mesh.hip
    make -C metadynamics-plugin_amd/csrc -s -j8 EXTRA_HIPFLAGS="$V" >> $LOG 2>&1 || { echo "build failed" | tee -a $LOG; continue; }
    if [ $round = 1 ]; then timeout -k 10 600 python3 -m pytest tests/test_gpu_mesh.py -x -q 2>&1 | tail -2 | tee -a $LOG; fi
    run "[$V]" MTD_EXP=1
    run "[$V]" MTD_EXP=1
    [ -n "$EXP_RIDER" ] && run "[$V] riders $EXP_RIDER" MTD_MESH_RIDER=$EXP_RIDER
  done
done
