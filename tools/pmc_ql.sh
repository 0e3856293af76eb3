#!/bin/bash
# Developer tool (GPU box): SQ counters of the two Steinhardt passes (tools/bench_ql.py), one counter set per rocprofv3 pass
# (--pmc with --kernel-trace only) -> gpurun_out/pmc_ql/summary.txt (per-launch medians per kernel)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_ql; rm -rf $O; mkdir -p $O
pass() { local tag=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$tag -o pmc -- python3 tools/bench_ql.py 12 > $O/$tag.log 2>&1
  cp "$(find $O/$tag -name '*counter_collection.csv' | head -1)" $O/${tag}.csv && rm -rf $O/$tag; }
pass a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
pass b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_IFETCH
pass c SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o ql -- python3 tools/bench_ql.py 60 > $O/kt.log 2>&1
cp "$(find $O/kt -name '*kernel_stats.csv' | head -1)" $O/kernel_stats.csv; rm -rf $O/kt
python3 - <<'PY' > gpurun_out/pmc_ql/summary.txt
import csv, collections, statistics, glob
out = collections.defaultdict(dict)
for f in sorted(glob.glob('gpurun_out/pmc_ql/[abc].csv')):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'k_ql_forces' in k or 'k_ql_accumulate' in k:
            acc[k.split('<')[0].split('::')[-1]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, cs in acc.items():
        for c, v in cs.items():
            out[k][c] = statistics.median(v)
for r in csv.DictReader(open('gpurun_out/pmc_ql/kernel_stats.csv')):
    for k in out:
        if k in r['Name']: out[k]['avg_us'] = float(r['AverageNs']) / 1e3
for k, cs in out.items():
    print(k)
    for c in sorted(cs): print("   %-28s %14.1f" % (c, cs[c]))
PY
cat gpurun_out/pmc_ql/summary.txt; grep "config 5" $O/kt.log
