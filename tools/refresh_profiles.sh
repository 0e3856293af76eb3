#!/bin/bash
# Developer tool (GPU box): re-collect everything under profiles/r2 that bench.py's line refers to, into gpurun_out/profiles_r2/
# (copy what is to be judged into profiles/r2 afterwards).  Every rocprofv3 run has the python program itself after `--`.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/profiles_r2; rm -rf $O; mkdir -p $O
set -x
timeout -k 10 400 python3 bench.py > $O/bench_default.log 2>&1 && tail -1 $O/bench_default.log > $O/bench_default.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o bench -- python3 bench.py --no-cpu-baseline --no-sub-records > $O/bench_under_rocprof.log 2>&1
grep '^{"metric"' $O/bench_under_rocprof.log | tail -1 > $O/bench_default_under_rocprof.json
cp $(find $O/kt -name "*kernel_stats.csv" | head -1) $O/bench_default_kernel_stats.csv
cp $(find $O/kt -name "*agent_info.csv" | head -1) $O/agent_info.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o pmc -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-sub-records --driver abi > $O/pmc_$c.log 2>&1
  cp $(find $O/pmc_$c -name "*counter_collection.csv" | head -1) $O/pmc_${c}_counter_collection.csv
done
python3 tools/pmc_summary.py $O/pmc_FETCH_SIZE_counter_collection.csv $O/pmc_WRITE_SIZE_counter_collection.csv $O/pmc_summary.json > /dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -o mesh -- python3 tools/bench_mesh.py 60 > $O/config3.log 2>&1
cp $(find $O/c3 -name "*kernel_stats.csv" | head -1) $O/config3_mesh_kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5 -o ql -- python3 tools/bench_ql.py 60 > $O/config5.log 2>&1
cp $(find $O/c5 -name "*kernel_stats.csv" | head -1) $O/config5_steinhardt_kernel_stats.csv
# the particle-sharded code path with one rank (mailbox to itself) and with two ranks sharing this GPU (rehearsal: software path only)
MTD_BENCH_FORCE_DIST=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-sub-records > $O/mailbox_1rank.log 2>&1; grep '^{"metric"' $O/mailbox_1rank.log | tail -1 > $O/bench_mailbox_1rank.json
MTD_XGMI_MAILBOX=0 MTD_BENCH_FORCE_DIST=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-sub-records > $O/rccl_1rank.log 2>&1; grep '^{"metric"' $O/rccl_1rank.log | tail -1 > $O/bench_rccl_1rank.json
MTD_BENCH_REHEARSAL=1 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29561 bench.py --gpus 2 --particles 500000 > $O/rehearsal2.log 2>&1; grep '^{"metric"' $O/rehearsal2.log | tail -1 > $O/bench_rehearsal_2ranks_one_gpu.json
set +x
rm -rf $O/kt $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/c3 $O/c5
ls -la $O; cat $O/pmc_summary.json | head -20; grep "config" $O/config3.log $O/config5.log
