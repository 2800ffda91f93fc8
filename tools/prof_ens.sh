#!/bin/bash
# kernel-trace of NN_Ens.fit at a reduced cfg4 shape (128 members, run on the GPU box from the repo root)
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_ens
mkdir -p $out
cat > /tmp/run_ens.py <<'PY'
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from quinn_amd.nns.mlp import MLP
from quinn_amd.solvers.nn_ens import NN_Ens
rs = np.random.RandomState(0)
N = 16384
x = rs.rand(N, 1) * 2 * np.pi - np.pi
y = np.sin(x) + 0.02 * rs.randn(N, 1)
ens = NN_Ens(MLP(1, 1, (256, 256, 256, 256), activ='tanh'), nens=128, dfrac=0.8)
ens.fit(x, y, val=[x[:2048], y[:2048]], lrate=0.01, nepochs=1, perm_mode='device', freq_out=1000)
torch.cuda.synchronize(); t0 = time.perf_counter()
ens.fit(x, y, val=[x[:2048], y[:2048]], lrate=0.01, nepochs=4, perm_mode='device', freq_out=1000)
torch.cuda.synchronize(); print("total for 4 steps", time.perf_counter() - t0)
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 /tmp/run_ens.py > $out/trace.log 2>&1
tail -2 $out/trace.log
python3 tools/prof_summary.py $out | head -28
