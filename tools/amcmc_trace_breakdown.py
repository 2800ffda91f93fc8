#!/usr/bin/env python3
"""Where the adapted phase of the device AMCMC engine spends its time: from a rocprofv3 kernel trace of tools/bench_amcmc_device.py
(tools/prof_amcmc.sh), per kernel name the summed duration, count and per-step share inside the window between the N-th and the last
history-product launch, the union of all kernels' busy intervals (the GPU's busy time) and the wall time of the window.
usage: amcmc_trace_breakdown.py <trace dir> [first history-product launch to count from, default 40]"""
import csv, glob, sys, collections
d = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 40
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=lambda p: -__import__("os").path.getmtime(p))[0]
rows = []
with open(f) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
hist = [r for r in rows if "k_hist_block16" in r[2]]
if len(hist) <= skip + 2:
    sys.exit("not enough history-product launches in the trace")
t0, t1 = hist[skip][0], hist[-1][0]
win = [r for r in rows if t0 <= r[0] < t1]
nacc = sum(1 for r in win if "k_accept" in r[2])
ngroups = 2 if nacc else 1
tot = collections.defaultdict(lambda: [0, 0])
for s, e, n in win:
    key = n.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")[:60]
    tot[key][0] += e - s; tot[key][1] += 1
busy, cur_s, cur_e = 0, None, None
for s, e, n in win:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
wall = t1 - t0
fwd = sum(c for k, (t, c) in tot.items() if "k_fused_fwd_i8" in k)
print(f"window: {wall / 1e6:.1f} ms, {fwd} forward launches, {nacc} accept launches; GPU busy (union of kernels) {busy / 1e6:.1f} ms = {busy / wall:.3f}")
print(f"{'kernel':60s} {'launches':>9s} {'sum ms':>9s} {'share of wall':>14s} {'avg us':>8s}")
for k, (t, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:18]:
    print(f"{k:60s} {c:9d} {t / 1e6:9.2f} {t / wall:14.3f} {t / c / 1e3:8.2f}")
