#!/usr/bin/env python3
"""Developer tool: per-kernel duration and the idle gap in FRONT of each kernel from a rocprofv3 kernel trace (csv), for the
launches of a short burst and of a long run separately: trace_gaps.py <kernel_trace.csv> [n_tail]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
n_tail = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def table(sel, title):
    dur, gap, cnt = collections.Counter(), collections.Counter(), collections.Counter()
    prev_end = None
    for r in sel:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("<")[0].split("(")[0]
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        dur[name] += e - s
        cnt[name] += 1
        if prev_end is not None and s - prev_end < 200000:
            gap[name] += max(0, s - prev_end)
        prev_end = e
    span = int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])
    print("%s: %d launches over %.1f us" % (title, len(sel), span / 1e3))
    for n, c in cnt.most_common():
        print("   %-40s n=%6d  dur %8.2f us   gap before %6.2f us" % (n[:40], c, dur[n] / c / 1e3, gap[n] / c / 1e3))
    print("   busy %.1f %%" % (100.0 * sum(dur.values()) / span))


table(rows[-n_tail:], "tail")
mid = len(rows) // 3
table(rows[mid:mid + n_tail], "middle third")
