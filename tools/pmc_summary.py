#!/usr/bin/env python3
"""Developer tool: per-kernel figures from rocprofv3 counter passes (each collected in a run of its own with --kernel-trace only, as
MI355X_MICROARCH.md prescribes).
  pmc_summary.py fused <fetch.csv> <write.csv> <kernel_stats.csv> <out.json>
        HBM bytes per launch of the two fused kernels + the kernel-trace average of the same command (bench.py reads this file)
  pmc_summary.py kernels <dir with pmc_*_counter_collection.csv and config*_kernel_stats.csv> <out.json>
        every mesh / Steinhardt kernel: HBM bytes per launch, kernel-trace average duration, achieved GB/s, SQ counters
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE counts 64 B per 128-byte request (the guide's correction)."""
import collections, csv, glob, json, os, re, sys
import numpy as np


def short(name):
    m = re.search(r"\b(k_[a-z0-9_]+)", name)
    return m.group(1) if m else name[:40]


def counters(path):
    """{kernel: {counter: [values per launch]}}"""
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        out[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


def stats(path):
    return {short(r["Name"]): (float(r["AverageNs"]) / 1e3, int(r["Calls"])) for r in csv.DictReader(open(path))}


if sys.argv[1] == "fused":
    fetch, write, st = counters(sys.argv[2]), counters(sys.argv[3]), stats(sys.argv[4])
    out = {}
    for k in ("k_fused_cv", "k_fused_force"):
        f, w = float(np.median(fetch[k]["FETCH_SIZE"])), float(np.median(write[k]["WRITE_SIZE"]))
        out[k] = {"FETCH_SIZE_KB_median": f, "WRITE_SIZE_KB_median": w, "hbm_read_bytes_per_launch": 2 * f * 1024, "hbm_write_bytes_per_launch": w * 1024,
                  "hbm_bytes_per_launch": (2 * f + w) * 1024, "launches": len(fetch[k]["FETCH_SIZE"]),
                  "rocprof_avg_launch_us": st.get(k, (None, 0))[0], "rocprof_calls": st.get(k, (None, 0))[1]}
    out["_note"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (each with --kernel-trace only) over "
                    "`bench.py --steps 200 --warmup 20 --no-cpu-baseline --driver abi`; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per "
                    "MI355X_MICROARCH.md (gfx950 FETCH_SIZE counts 64 B per 128-B request); rocprof_avg_launch_us: --kernel-trace --stats "
                    "of `bench.py` (default 2000 steps)")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    out["kernel_source_sha256"] = bench.kernel_source_sha()      # bench.py reports the traffic only for these kernel sources
    json.dump(out, open(sys.argv[5], "w"), indent=1)
    print(json.dumps(out, indent=1))
else:
    d, out_path = sys.argv[2], sys.argv[3]
    out = {}
    for tag, stat_file in (("mesh", "config3_mesh_kernel_stats.csv"), ("ql", "config5_steinhardt_kernel_stats.csv")):
        sp = os.path.join(d, stat_file)
        if not os.path.exists(sp):
            continue
        st = stats(sp)
        per = collections.defaultdict(dict)
        for f in glob.glob(os.path.join(d, "pmc_%s_*_counter_collection.csv" % tag)):
            for k, cs in counters(f).items():
                for c, v in cs.items():
                    per[k][c] = float(np.median(v))
        for k, cs in per.items():
            if not k.startswith("k_") or k not in st:
                continue
            us = st[k][0]
            rec = {"rocprof_avg_launch_us": us, "calls": st[k][1]}
            if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
                rd, wr = 2 * cs["FETCH_SIZE"] * 1024, cs["WRITE_SIZE"] * 1024
                rec.update(hbm_read_bytes_per_launch=rd, hbm_write_bytes_per_launch=wr, hbm_bytes_per_launch=rd + wr,
                           achieved_GBs=(rd + wr) / (us * 1e-6) / 1e9, frac_of_8TBs=(rd + wr) / (us * 1e-6) / 8e12)
            for c, v in cs.items():
                if c.startswith("SQ_"):
                    rec[c] = v
            if "SQ_INSTS_VALU_FMA_F64" in cs:
                # fp64 flops of the launch as issued: wave instructions x 64 lanes (an upper bound: every lane counted as active)
                rec["fp64_flops_per_launch"] = 64.0 * (2.0 * cs["SQ_INSTS_VALU_FMA_F64"] + cs.get("SQ_INSTS_VALU_MUL_F64", 0.0) + cs.get("SQ_INSTS_VALU_ADD_F64", 0.0))
            if "SQ_ACTIVE_INST_VALU" in cs and "SQ_BUSY_CYCLES" in cs and cs["SQ_BUSY_CYCLES"] > 0:
                # SQ_ACTIVE_INST_VALU: cycles (x4, per SIMD) in which a VALU instruction is executing, summed over the chip;
                # SQ_BUSY_CYCLES: cycles the SQs were busy, summed over the shader engines — the ratio per SIMD follows the guide
                rec["valu_active_over_wave_cycles"] = cs["SQ_ACTIVE_INST_VALU"] / cs["SQ_WAVE_CYCLES"] if cs.get("SQ_WAVE_CYCLES") else None
            out[k] = rec
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    # bench.py reports these figures only while the kernels' sources are the ones the counters were collected on
    out["kernel_source_sha256"] = {"mesh": bench.kernel_source_sha("mesh"), "ql": bench.kernel_source_sha("ql")}
    out["_note"] = ("per launch, medians over the launches of tools/bench_mesh.py / tools/bench_ql.py under rocprofv3 --pmc (separate passes per "
                    "counter set, --kernel-trace only); durations from the --kernel-trace --stats run of the same program; "
                    "hbm bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 correction of MI355X_MICROARCH.md)")
    json.dump(out, open(out_path, "w"), indent=1)
    print("%-26s %9s %10s %10s %9s %8s" % ("kernel", "us", "read MB", "write MB", "GB/s", "of 8TB/s"))
    for k, r in sorted(out.items(), key=lambda kv: -(kv[1].get("rocprof_avg_launch_us", 0) if isinstance(kv[1], dict) else 0)):
        if isinstance(r, dict) and "hbm_bytes_per_launch" in r:
            print("%-26s %9.2f %10.2f %10.2f %9.0f %8.3f" % (k, r["rocprof_avg_launch_us"], r["hbm_read_bytes_per_launch"] / 1e6, r["hbm_write_bytes_per_launch"] / 1e6,
                                                          r["achieved_GBs"], r["frac_of_8TBs"]))
