#!/bin/bash
# Developer tool (GPU box): per-kernel table of config 5 (tools/bench_ql.py) under rocprofv3 -> gpurun_out/config5_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_ql
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ql -o ql -- python tools/bench_ql.py 60 > gpurun_out/prof_ql.log 2>&1
grep "config 5" gpurun_out/prof_ql.log
cp "$(find gpurun_out/prof_ql -name '*kernel_stats.csv' | head -1)" gpurun_out/config5_kernel_stats.csv
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/config5_kernel_stats.csv')):
    if float(r['Percentage']) > 0.5:
        print("%-58s %5s %9.1f us %6s%%" % (r['Name'].replace('(anonymous namespace)::','').replace('void ','')[:58], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
