#!/usr/bin/env python3
"""VALU-busy / MFMA-busy fractions of a kernel from a tools/prof.sh output directory (rocprofv3 --pmc passes):
valu_busy = 4 SQ_ACTIVE_INST_VALU / SIMD-cycles, mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / SIMD-cycles, SIMD-cycles = 1024 SIMDs x
GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs).  Updates profiles/pmc_busy.json[kernel].
usage: tools/pmc_busy.py <prof dir> <kernel-name-substring> <key> <source file name under profiles/>"""
import csv, glob, json, os, sys, collections
d, sub, key, src = sys.argv[1:5]
vals = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in vals.items()}
simd_cycles = 1024 * m["GRBM_GUI_ACTIVE"] / 8
out = {"valu_busy": 4 * m["SQ_ACTIVE_INST_VALU"] / simd_cycles, "mfma_busy": m["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles,
       "insts_valu_per_launch": m.get("SQ_INSTS_VALU"), "insts_mfma_per_launch": m.get("SQ_INSTS_MFMA"),
       "lds_bank_conflict_frac": m.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, m.get("SQ_LDS_IDX_ACTIVE", 1)),
       "wait_any_frac_of_wave_cycles": m.get("SQ_WAIT_ANY", 0) / max(1.0, m.get("SQ_WAVE_CYCLES", 1)), "profile": "profiles/" + src}
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(root, "gpurun_out", "pmc_busy.json")
allv = json.load(open(path)) if os.path.exists(path) else {}
allv[key] = out
json.dump(allv, open(path, "w"), indent=1)
print(key, json.dumps(out))
