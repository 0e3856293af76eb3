#!/bin/bash
# Developer tool (GPU box): re-collect everything under profiles/<round> that bench.py's line refers to, into gpurun_out/profiles_<round>/
# (copy what is to be judged into profiles/<round> afterwards; ROUND=r4 unless set).  Every rocprofv3 run has the python program itself after `--`;
# counters are collected in runs of their own (--pmc with --kernel-trace only), one counter set per pass.
# usage: tools/refresh_profiles.sh [bench|mesh|ql|sharded ...]   (default: all)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ROUND=${ROUND:-r4}
O=gpurun_out/profiles_$ROUND; mkdir -p $O
WHAT="${*:-bench mesh ql sharded}"
# (one summary file carries the mesh and the Steinhardt kernels and a box starts empty: the two are always collected together)
if [[ " $WHAT " == *" mesh "* && " $WHAT " != *" ql "* ]]; then WHAT="$WHAT ql"; fi
if [[ " $WHAT " == *" ql "* && " $WHAT " != *" mesh "* ]]; then WHAT="$WHAT mesh"; fi
rocprofv3 -L > $O/counters_list.txt 2>&1 || true
pmc_pass() {   # pmc_pass <tag> <counters...> -- <program...>
  local tag=$1; shift; local ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  rm -rf $O/pmc_$tag
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc "${ctr[@]}" --output-format csv -d $O/pmc_$tag -o pmc -- "$@" > $O/pmc_$tag.log 2>&1
  local f="$(find $O/pmc_$tag -name '*counter_collection.csv' | head -1)"
  if [ -n "$f" ]; then cp "$f" $O/pmc_${tag}_counter_collection.csv; rm -rf $O/pmc_$tag; else echo "pmc pass $tag produced no counters (see $O/pmc_$tag.log)"; fi
}
set -x
if [[ " $WHAT " == *" bench "* ]]; then
  # counters of the headline kernels first: the bench lines below read profiles/$ROUND/pmc_summary.json and report `traffic` only
  # while its hash of the kernel sources matches
  for c in FETCH_SIZE WRITE_SIZE; do
    pmc_pass $c $c -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-sub-records --no-variants --driver abi
  done
  rm -rf $O/kt
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o bench -- python3 bench.py --no-cpu-baseline --no-sub-records --no-variants > $O/bench_under_rocprof.log 2>&1
  grep '^{"metric"' $O/bench_under_rocprof.log | tail -1 > $O/bench_default_under_rocprof.json
  cp "$(find $O/kt -name '*kernel_stats.csv' | head -1)" $O/bench_default_kernel_stats.csv
  cp "$(find $O/kt -name '*agent_info.csv' | head -1)" $O/agent_info.csv; rm -rf $O/kt
  python3 tools/pmc_summary.py fused $O/pmc_FETCH_SIZE_counter_collection.csv $O/pmc_WRITE_SIZE_counter_collection.csv $O/bench_default_kernel_stats.csv $O/pmc_summary.json > /dev/null
  mkdir -p profiles/$ROUND && cp $O/pmc_summary.json $O/bench_default_kernel_stats.csv profiles/$ROUND/
fi
if [[ " $WHAT " == *" mesh "* ]]; then
  rm -rf $O/c3
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -o mesh -- python3 tools/bench_mesh.py 60 > $O/config3.log 2>&1
  cp "$(find $O/c3 -name '*kernel_stats.csv' | head -1)" $O/config3_mesh_kernel_stats.csv; rm -rf $O/c3
  for c in FETCH_SIZE WRITE_SIZE; do pmc_pass mesh_$c $c -- python3 tools/bench_mesh.py 20; done
  pmc_pass mesh_SQ SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -- python3 tools/bench_mesh.py 12
  pmc_pass mesh_SQ2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -- python3 tools/bench_mesh.py 12
fi
if [[ " $WHAT " == *" ql "* ]]; then
  rm -rf $O/c5
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5 -o ql -- python3 tools/bench_ql.py 60 > $O/config5.log 2>&1
  cp "$(find $O/c5 -name '*kernel_stats.csv' | head -1)" $O/config5_steinhardt_kernel_stats.csv; rm -rf $O/c5
  for c in FETCH_SIZE WRITE_SIZE; do pmc_pass ql_$c $c -- python3 tools/bench_ql.py 20; done
  pmc_pass ql_SQ SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -- python3 tools/bench_ql.py 12
  pmc_pass ql_SQ2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -- python3 tools/bench_ql.py 12
  # the fp64 arithmetic actually issued (bench.py prices config 5's roofline with these: 2 FMA + MUL + ADD, 64 lanes each)
  pmc_pass ql_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 -- python3 tools/bench_ql.py 12
fi
if [[ " $WHAT " == *" mesh "* || " $WHAT " == *" ql "* ]]; then
  python3 tools/pmc_summary.py kernels $O $O/pmc_mesh_ql_summary.json > $O/pmc_mesh_ql_summary.txt
  mkdir -p profiles/$ROUND && cp $O/pmc_mesh_ql_summary.json profiles/$ROUND/
fi
if [[ " $WHAT " == *" bench "* ]]; then
  # the lines themselves LAST: they read the counter summaries written above (profiles/$ROUND/ of this copy of the tree)
  timeout -k 10 600 python3 bench.py > $O/bench_default.log 2>&1 && tail -1 $O/bench_default.log > $O/bench_default.json
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $O/bench_k20.log 2>&1 && tail -1 $O/bench_k20.log > $O/bench_driver_call_k20.json
fi
if [[ " $WHAT " == *" sharded "* ]]; then
  # the particle-sharded code path with one rank (mailbox to itself) and with two ranks sharing this GPU (rehearsal: software path only)
  MTD_BENCH_FORCE_DIST=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-sub-records --no-variants > $O/mailbox_1rank.log 2>&1; grep '^{"metric"' $O/mailbox_1rank.log | tail -1 > $O/bench_mailbox_1rank.json
  MTD_XGMI_MAILBOX=0 MTD_BENCH_FORCE_DIST=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-sub-records --no-variants > $O/rccl_1rank.log 2>&1; grep '^{"metric"' $O/rccl_1rank.log | tail -1 > $O/bench_rccl_1rank.json
  timeout -k 10 400 python3 bench.py --gpus 2 --particles 500000 > $O/rehearsal2.log 2>&1; grep '^{"metric"' $O/rehearsal2.log | tail -1 > $O/bench_rehearsal_2ranks_weak.json
  timeout -k 10 400 python3 bench.py --gpus 2 --scaling strong --particles 1000000 > $O/rehearsal2s.log 2>&1; grep '^{"metric"' $O/rehearsal2s.log | tail -1 > $O/bench_rehearsal_2ranks_strong.json
  timeout -k 10 400 python3 bench.py --gpus 2 --walkers --particles 500000 --steps 500 --warmup 50 > $O/rehearsal2w.log 2>&1; grep '^{"metric"' $O/rehearsal2w.log | tail -1 > $O/bench_rehearsal_2walkers.json
  # configs 3 and 5 domain decomposed inside the C++ host classes (two processes on this GPU: the code path is real, the numbers mean nothing)
  timeout -k 10 400 python3 bench.py --config 3 --gpus 2 --particles 500000 --steps 200 --warmup 20 > $O/rehearsal_c3r.log 2>&1; grep '^{"metric"' $O/rehearsal_c3r.log | tail -1 > $O/bench_rehearsal_config3_replicated.json
  timeout -k 10 400 python3 bench.py --config 3 --gpus 2 --particles 500000 --steps 200 --warmup 20 --mesh slab > $O/rehearsal_c3s.log 2>&1; grep '^{"metric"' $O/rehearsal_c3s.log | tail -1 > $O/bench_rehearsal_config3_slab.json
  timeout -k 10 400 python3 bench.py --config 5 --gpus 2 --steps 200 --warmup 20 > $O/rehearsal_c5.log 2>&1; grep '^{"metric"' $O/rehearsal_c5.log | tail -1 > $O/bench_rehearsal_config5.json
  # the same entry at N = 1 (the single-GPU lines of configs 3 and 5 through bench.py --config)
  timeout -k 10 400 python3 bench.py --config 3 > $O/config3_n1.log 2>&1; grep '^{"metric"' $O/config3_n1.log | tail -1 > $O/bench_config3_n1.json
  timeout -k 10 400 python3 bench.py --config 5 > $O/config5_n1.log 2>&1; grep '^{"metric"' $O/config5_n1.log | tail -1 > $O/bench_config5_n1.json
fi
set +x
ls -la $O
