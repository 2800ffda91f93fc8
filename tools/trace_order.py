#!/usr/bin/env python3
"""From a rocprofv3 kernel trace: durations of one kernel in dispatch order, and the gaps between consecutive dispatches
(end -> next start), summarised per block of 50.  usage: tools/trace_order.py <dir> <name-substring>"""
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
name = sys.argv[2]
prev_end = None
dur, gap = [], []
for s, e, k in sel:
    if name in k:
        dur.append((e - s) / 1e3)
        gap.append((s - prev_end) / 1e3 if prev_end else 0.0)
    prev_end = e
print("dispatches", len(dur))
for i in range(0, len(dur), 50):
    d, g = dur[i:i + 50], gap[i + 1:i + 50]
    if len(d) < 10: continue
    print("block %3d: dur avg %.2f min %.2f max %.2f us | gap to previous kernel avg %.2f us" % (i // 50, sum(d) / len(d), min(d), max(d), sum(g) / max(1, len(g))))
