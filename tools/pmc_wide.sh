#!/bin/bash
# PMC pass over the wide int8-slice kernels at the cfg4 / cfg3 shapes: tools/pmc_wide.sh "<counters>" [kernel-substring]
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_wide
rm -rf $out; mkdir -p $out
[ -f /tmp/run_wide.py ] || sed -n '/^cat > \/tmp\/run_wide.py/,/^PY$/p' tools/ab_wide.sh | sed '1d;$d' > /tmp/run_wide.py
rocprofv3 --pmc $1 --output-format csv -d $out -- python3 /tmp/run_wide.py > $out/log.txt 2>&1
python3 - "$out" "${2:-k_i8_}" <<'PY'
import csv, glob, os, sys, collections
out, pat = sys.argv[1], sys.argv[2]
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        if pat not in k: continue
        print("==", k)
        for c, v in sorted(cs.items()):
            print("     %-28s avg/dispatch=%.4g min=%.4g max=%.4g (n=%d)" % (c, sum(v) / len(v), min(v), max(v), len(v)))
PY
