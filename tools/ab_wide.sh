#!/bin/bash
# kernel times of the wide int8-slice kernels for several library builds (tools/ab_build.sh <name> -D...), one gpurun call:
#   tools/ab_wide.sh <name|base> ...      (GPU box, repo root)
export TMPDIR=/tmp
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
for v in "$@"; do
  if [ $v == base ]; then unset QUINN_AMD_LIB; else export QUINN_AMD_LIB=$PWD/quinn_amd/lib/libquinn_amd_$v.so; fi
  out=$PWD/gpurun_out/ab_wide_$v
  rm -rf $out; mkdir -p $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 /tmp/run_wide.py > $out/trace.log 2>&1
  echo "== $v"
  python3 tools/prof_summary.py $out | grep -A30 "== kernel trace" | grep "k_i8_wide\|k_i8_slice\|k_gemm64\|k_dW" | cut -c1-160
done
