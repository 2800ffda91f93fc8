#!/bin/bash
# Round-2 profile of the int8-slice kernels for wide networks (qn_wide_i8.hip, qn_dw_i8.hip) at the cfg4 (4x256, N = 16384,
# 32 members) and cfg3 (3x128, N = 8192, 128 samples) shapes: rocprofv3 --kernel-trace --stats, then three PMC passes.
# GPU box, repo root; the profiled program goes directly after `--`.
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_wide
rm -rf $out; mkdir -p $out
cat > /tmp/run_wide.py <<'PY'
import sys, os, numpy as np, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from quinn_amd.ops import MLPArch, BatchedMLP
for dims, N, B in (((1, 256, 256, 256, 256, 1), 16384, 32), ((2, 128, 128, 128, 1), 8192, 128)):
    arch = MLPArch(dims, "tanh"); rs = np.random.RandomState(0)
    x = rs.rand(N, dims[0]) * 6 - 3; y = np.sin(x).sum(axis=1, keepdims=True)
    op = BatchedMLP(arch, x, y); W = op.weights(0.1 * rs.randn(B, arch.nparams))
    for _ in range(3): op.sse_grad(W)
    for _ in range(3): op.sse(W)
    torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 /tmp/run_wide.py > $out/trace.log 2>&1
python3 tools/prof_summary.py $out | grep -v "^== pmc" | head -40
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE GRBM_COUNT TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d $out/pmc_$n -- python3 /tmp/run_wide.py > $out/pmc_$n.log 2>&1 || echo "pmc $n failed"
done
python3 - <<'PY'
import csv, glob, os, collections
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "prof_wide")
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d): continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:75]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            if "k_i8_wide" not in k and "k_i8_dw" not in k: continue
            print("==", os.path.basename(d), k)
            for c, v in sorted(cs.items()):
                print("     %-28s avg/dispatch=%.4g (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
