#!/bin/bash
# Developer tool (GPU box): per-kernel table of config 3 (tools/bench_mesh.py) under rocprofv3 -> gpurun_out/config3_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_mesh
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_mesh -o mesh -- python tools/bench_mesh.py 60 ${1:-} > gpurun_out/prof_mesh.log 2>&1
grep "config 3" gpurun_out/prof_mesh.log
f=$(find gpurun_out/prof_mesh -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/config3_kernel_stats.csv
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/config3_kernel_stats.csv')):
    if float(r['Percentage']) > 0.5:
        print("%-58s %5s %9.1f us %6s%%" % (r['Name'].replace('(anonymous namespace)::','').replace('void ','')[:58], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
