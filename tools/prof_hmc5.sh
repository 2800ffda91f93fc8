#!/bin/bash
# kernel trace of device HMC at cfg5 (256 chains, 4x256, N = 32768, L = 10): 1 warm-up step + 2 timed steps
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_hmc5
rm -rf $out; mkdir -p $out
cat > /tmp/run_hmc5.py <<'PY'
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from quinn_amd.ops import MLPArch, BatchedMLP
from quinn_amd.mcmc.device_hmc import DeviceHMC
dims, N, C, L = (1, 256, 256, 256, 256, 1), 32768, 256, 10
arch = MLPArch(dims, "tanh"); rs = np.random.RandomState(0)
x = rs.rand(N, 1) * 2 * np.pi - np.pi; y = 0.02 * rs.randn(N, 1) + np.sin(x)
op = BatchedMLP(arch, x, y)
ini = np.stack([0.1 * np.random.RandomState(1000 + c).randn(arch.nparams) for c in range(C)])
eng = DeviceHMC(op, 0.02, epsilon=float(os.environ.get("QN_EPS", "0.0001")), L=L, seed=1)
eng.run(1, ini, store_chain=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
r = eng.run(2, ini, store_chain=False)
torch.cuda.synchronize(); el = time.perf_counter() - t0
print("accrate", float(r["accrate"].mean()), "wall per step", el / 2, "grad TFLOP/s (20 evals)", 20 * C * arch.flops_fwdbwd(N) / el / 1e12, "(21 evals)", 21 * C * arch.flops_fwdbwd(N) / el / 1e12)
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 /tmp/run_hmc5.py > $out/trace.log 2>&1
grep wall $out/trace.log
python3 tools/prof_summary.py $out | grep -A16 "== kernel stats" | cut -c1-170
