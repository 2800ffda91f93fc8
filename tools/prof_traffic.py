#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE passes of tools/prof_traffic.sh into profiles-ready JSON.
gfx950: FETCH_SIZE (KiB) reports half of a wide coalesced read stream -> doubled (MI355X_MICROARCH.md
section HBM); WRITE_SIZE (KiB) is exact.  Bytes are summed over all kernels of one bench step."""
import csv, glob, json, os, sys, collections
out = sys.argv[1]
res = {}
for kind in ("logpost", "grad"):
    tot = {}
    detail = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        per_kernel = collections.defaultdict(list)
        for f in glob.glob(os.path.join(out, f"{kind}_{c}", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == c:
                    per_kernel[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
        step = 0.0
        for k, v in per_kernel.items():
            if "fused" in k or "k_sum" in k or "k_grad" in k or "k_fwd" in k or "k_dW" in k or "k_bwd" in k or "k_sse" in k:
                avg = sum(v) / len(v)
                detail[f"{c}:{k}"] = avg
                step += avg
        tot[c] = step
    kib = 2.0 * tot.get("FETCH_SIZE", 0.0) + tot.get("WRITE_SIZE", 0.0)
    # key = bench.py's config.kernel_path: the sliced int8-product kernels or the float64-MFMA ones
    fam = "fused_i8" if any("fwd_i8" in k for k in detail) else "fused_i8_bwd" if any("bwd_i8" in k for k in detail) else \
        ("fused_dp_bwd" if kind == "grad" else "fused_dp")
    res[f"{kind}_f64_{fam}"] = {"hbm_bytes_per_launch": kib * 1024.0, "fetch_kib_raw": tot.get("FETCH_SIZE"),
                                "write_kib": tot.get("WRITE_SIZE"), "detail_kib": detail, "round": int(os.environ.get("QN_ROUND", "4"))}
json.dump(res, open(os.path.join(out, "hbm_traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
