#!/usr/bin/env python3
"""Condense rocprofv3 csv output (kernel trace stats + pmc passes) into one text summary."""
import csv, glob, os, sys, collections
out = sys.argv[1]
lines = []
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    lines.append("== kernel stats (%s)" % os.path.basename(f))
    for r in csv.DictReader(open(f)):
        lines.append("  %-90s calls=%s total_ns=%s avg_ns=%s pct=%s" % (r.get("Name", "")[:90], r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage")))
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(list)
    meta = {}
    for r in rows:
        k = r["Kernel_Name"][:90]
        agg[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        meta[k] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size"), r.get("Workgroup_Size"), r.get("Grid_Size"))
    lines.append("== kernel trace")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        v2 = sorted(v)
        lines.append("  %-90s n=%d avg_us=%.2f med_us=%.2f min_us=%.2f vgpr/agpr/sgpr/lds/scratch/wg/grid=%s" % (k, len(v), sum(v) / len(v) / 1e3, v2[len(v2) // 2] / 1e3, v2[0] / 1e3, meta[k]))
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        lines.append("== pmc %s" % os.path.basename(d))
        for k, cs in agg.items():
            if "fused" not in k and "k_fwd" not in k and "k_dW" not in k and "k_bwd" not in k:
                continue
            lines.append("  " + k)
            for c, v in sorted(cs.items()):
                lines.append("     %-28s avg/dispatch=%.4g (n=%d)" % (c, sum(v) / len(v), len(v)))
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
