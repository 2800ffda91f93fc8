#!/usr/bin/env python3
"""End-to-end AMCMC at the headline configuration (64 chains, 3x64 tanh MLP, p=8513, N=4096) on the
device-resident engine: steps/s before the first adaptation (structured initial proposal), cost of
an adaptation (windowed SYRK + batched Cholesky), steps/s with full p x p proposal factors."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quinn_amd.ops import MLPArch, BatchedMLP
from quinn_amd.mcmc.device_amcmc import DeviceAMCMC

C, N = 64, 4096
arch = MLPArch((1, 64, 64, 64, 1), "tanh")
rs = np.random.RandomState(0)
x = rs.rand(N, 1) * 2 * np.pi - np.pi
y = 0.02 * rs.randn(N, 1) + np.sin(x)
op = BatchedMLP(arch, x, y)
ini = np.stack([np.random.RandomState(1000 + c).rand(arch.nparams) for c in range(C)])
fdt = torch.float32 if len(sys.argv) > 1 and sys.argv[1] == "f32" else torch.float64
tadapt = int(os.environ.get("TADAPT", "300"))
res = {}
# phase A: no adaptation inside the run
UG = os.environ.get("USE_GRAPH", "1") == "1"
eng = DeviceAMCMC(op, 0.02, gamma=0.01, t0=100, tadapt=1000, seed=1, use_graph=UG)      # 300 steps: no adaptation yet
eng.run(20, ini, store_chain=True)                      # warm-up (allocator, RNG, first launches)
torch.cuda.synchronize(); t0 = time.perf_counter()
r = eng.run(300, ini, store_chain=True)
torch.cuda.synchronize(); ta = time.perf_counter() - t0
res["phaseA_steps_per_s"] = 300 / ta
res["phaseA_logpost_evals_per_s"] = 300 * C / ta
res["phaseA_accrate"] = float(r["accrate"].mean())
del r, eng
torch.cuda.empty_cache()
# phase B: adaptation at step tadapt, then draws through the full factor
eng = DeviceAMCMC(op, 0.02, gamma=0.01, t0=100, tadapt=tadapt, seed=1, factor_dtype=fdt, chol_chunk=4, use_graph=UG)
n1 = tadapt + 1
eng.run(n1, ini, store_chain=False)                     # warm-up incl. rocSOLVER / rocBLAS handles
eng = DeviceAMCMC(op, 0.02, gamma=0.01, t0=100, tadapt=tadapt, seed=1, factor_dtype=fdt, chol_chunk=4, use_graph=UG)
torch.cuda.synchronize(); t0 = time.perf_counter()
r = eng.run(n1, ini, store_chain=False)
torch.cuda.synchronize(); t1 = time.perf_counter() - t0
eng2 = DeviceAMCMC(op, 0.02, gamma=0.01, t0=100, tadapt=tadapt, seed=1, factor_dtype=fdt, chol_chunk=4, use_graph=UG)
n2 = tadapt + 41
torch.cuda.synchronize(); t0 = time.perf_counter()
r2 = eng2.run(n2, ini, store_chain=False)
torch.cuda.synchronize(); t2 = time.perf_counter() - t0
res["factor_dtype"] = str(fdt)
res["use_graph"] = UG
res["tadapt"] = tadapt
res["adaptation_seconds_incl_%d_steps" % n1] = t1
res["phaseB_ms_per_step"] = 1e3 * (t2 - t1) / (n2 - n1)
res["phaseB_steps_per_s"] = (n2 - n1) / (t2 - t1)
res["phaseB_accrate_overall"] = float(r2["accrate"].mean())
res["peak_mem_GB"] = torch.cuda.max_memory_allocated() / 1e9
print(json.dumps(res))
