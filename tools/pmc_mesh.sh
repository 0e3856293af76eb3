#!/bin/bash
# Developer tool (GPU box): SQ counters per kernel of config 3 (tools/bench_mesh.py) -> gpurun_out/pmc_mesh_summary.txt
# (counters in their own run: --pmc with --kernel-trace only)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_mesh
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS \
    --kernel-trace --output-format csv -d gpurun_out/pmc_mesh -o mesh -- python3 tools/bench_mesh.py 12 ${1:-} > gpurun_out/pmc_mesh.log 2>&1
python3 - <<'PY' | tee gpurun_out/pmc_mesh_summary.txt
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_mesh/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]; i = n.find("k_"); n = n[i:i + 24] if i >= 0 else n[:24]
    acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": calls[n] += 1
names = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT"]
print("%-26s %5s " % ("kernel (per launch)", "n") + " ".join("%12s" % x[3:15] for x in names))
for n, c in acc.items():
    if calls[n]: print("%-26s %5d " % (n, calls[n]) + " ".join("%12.3g" % (c[x] / calls[n]) for x in names))
PY
