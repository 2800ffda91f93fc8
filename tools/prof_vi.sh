#!/bin/bash
# kernel-trace of NN_VI.fit at the cfg3 shape (run on the GPU box from the repo root)
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_vi
mkdir -p $out
cat > /tmp/run_vi.py <<'PY'
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from quinn_amd.nns.mlp import MLP
from quinn_amd.solvers.nn_vi import NN_VI
rs = np.random.RandomState(0)
N = 8192
x = rs.rand(N, 2) * 2 * np.pi - np.pi
y = np.sin(x).sum(axis=1, keepdims=True) + 0.02 * rs.randn(N, 1)
torch.manual_seed(0)
vi = NN_VI(MLP(2, 1, (128, 128, 128), activ='tanh'), rng='device')
vi.fit(x, y, val=[x[:1024], y[:1024]], datanoise=0.02, lrate=0.01, nsam=128, nepochs=2, freq_out=1000)
torch.cuda.synchronize(); t0 = time.perf_counter()
vi.fit(x, y, val=[x[:1024], y[:1024]], datanoise=0.02, lrate=0.01, nsam=128, nepochs=20, freq_out=1000)
torch.cuda.synchronize(); print("per step", (time.perf_counter() - t0) / 20)
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 /tmp/run_vi.py > $out/trace.log 2>&1
tail -2 $out/trace.log
python3 tools/prof_summary.py $out | head -40
