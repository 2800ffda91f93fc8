#!/bin/bash
# kernel trace + counters of the streaming fused forward at the cfg3 shape (run on the GPU box from the repo root)
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_stream
mkdir -p $out
cat > /tmp/run_stream.py <<'PY'
import sys, os, numpy as np, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from quinn_amd.ops import MLPArch, BatchedMLP
dims, N, B = (2, 128, 128, 128, 1), 8192, 128
arch = MLPArch(dims, "tanh"); rs = np.random.RandomState(0)
x = rs.rand(N, 2) * 6 - 3; y = np.sin(x).sum(axis=1, keepdims=True)
op = BatchedMLP(arch, x, y); W = op.weights(0.1 * rs.randn(B, arch.nparams))
for _ in range(10): op.sse(W)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 /tmp/run_stream.py > $out/trace.log 2>&1
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  n=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d $out/pmc_$n -- python3 /tmp/run_stream.py > $out/pmc_$n.log 2>&1 || echo "pmc $n failed"
done
python3 tools/prof_summary.py $out
