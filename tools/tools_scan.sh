#!/bin/bash
# developer tool: per-kernel durations (rocprofv3 kernel trace) of bench.py as a function of particle count
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  rm -rf $R/gpurun_out/scan_$n
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/scan_$n -- python3 $R/bench.py --steps 300 --warmup 50 --no-cpu-baseline --particles $n $BENCH_EXTRA > $R/gpurun_out/scan_$n.log 2>&1
  python3 - <<PY
import csv,glob,json
f=glob.glob("$R/gpurun_out/scan_$n/*/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
d=[json.loads(l) for l in open("$R/gpurun_out/scan_$n.log") if l.startswith("{")][-1]
print("N=%8d us/step=%6.2f  " % ($n, 1e3*d["ms_per_step"]) + "  ".join("%s=%.2f" % (r["Name"].split("::")[1].split("<")[0].split("(")[0], float(r["AverageNs"])/1e3) for r in rows[:2]))
PY
done
