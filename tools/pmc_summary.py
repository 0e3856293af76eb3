#!/usr/bin/env python3
"""Developer tool: HBM bytes per launch of the fused kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE collected
in separate runs with --kernel-trace only, as MI355X_MICROARCH.md prescribes).  usage: pmc_summary.py <fetch.csv> <write.csv> <out.json>
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE counts 64 B per 128-byte request (guide's correction)."""
import csv, json, sys
import numpy as np


def per_kernel(path, counter):
    vals = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        for key in ("k_fused_cv", "k_fused_force"):
            if key in r["Kernel_Name"]:
                vals.setdefault(key, []).append(float(r["Counter_Value"]))
    return vals


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in ("k_fused_cv", "k_fused_force"):
    f, w = float(np.median(fetch[k])), float(np.median(write[k]))
    out[k] = {"FETCH_SIZE_KB_median": f, "WRITE_SIZE_KB_median": w, "hbm_bytes_per_launch": (2 * f + w) * 1024, "launches": len(fetch[k])}
out["_note"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (each with --kernel-trace only) over "
                "`bench.py --steps 200 --warmup 20 --no-cpu-baseline --driver abi`; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per "
                "MI355X_MICROARCH.md (gfx950 FETCH_SIZE counts 64 B per 128-B request)")
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
out["kernel_source_sha256"] = bench.kernel_source_sha()      # bench.py reports the traffic only for these kernel sources
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
