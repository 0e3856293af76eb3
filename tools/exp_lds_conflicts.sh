#!/bin/bash
# Developer tool (GPU box): the LDS bank conflicts of k_tile_scatter (0.62 of its LDS cycles) and k_tile_forces (0.61) — are they on
# the critical path?  A DIAGNOSTIC build (-DMTD_EXP_LDS_NOCONFLICT: every lane's LDS accesses go to conflict-free addresses; the
# sums and forces are then WRONG) against the product build, per-kernel durations from rocprofv3's kernel trace.  Rebuilds the
# product library at the end.  Output: gpurun_out/r4/lds_conflicts.log
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4; mkdir -p $O; LOG=$O/lds_conflicts.log; : > $LOG
for V in "" "-DMTD_EXP_LDS_NOCONFLICT"; do
  touch metadynamics-plugin_amd/csrc/mesh.hip
  make -C metadynamics-plugin_amd/csrc -s -j8 EXTRA_HIPFLAGS="$V" >> $LOG 2>&1 || { echo "build failed [$V]" | tee -a $LOG; continue; }
  for r in 1 2; do
    rm -rf $O/prof_lds
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lds -o mesh -- python3 tools/bench_mesh.py 100 > $O/prof_lds.log 2>&1
    f=$(find $O/prof_lds -name "*kernel_stats.csv" | head -1)
    python3 - "$f" "[${V:-product build}]" <<'PY' | tee -a $LOG
import csv, re, sys
rows = {}
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"\b(k_[a-z0-9_]+)", r['Name'])
    if m: rows[m.group(1)] = float(r['AverageNs']) / 1e3
print(sys.argv[2], "k_tile_scatter %.2f us   k_tile_forces %.2f us   k_tile_bin %.2f us" % (rows.get('k_tile_scatter', 0), rows.get('k_tile_forces', 0), rows.get('k_tile_bin', 0)))
PY
    rm -rf $O/prof_lds
  done
done
touch metadynamics-plugin_amd/csrc/mesh.hip
make -C metadynamics-plugin_amd/csrc -s -j8 >> $LOG 2>&1
