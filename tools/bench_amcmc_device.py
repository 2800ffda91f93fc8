#!/usr/bin/env python3
"""End-to-end AMCMC at the headline configuration (64 chains, 3x64 tanh MLP, p=8513, N=4096) on the
device-resident engine: steps/s before the first adaptation (structured initial proposal) and after it
(adapted proposal drawn in sample space from the stored distinct states), over a long run with the
reference's defaults (t0=100, tadapt=1000) so that the history grows as it would in production."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quinn_amd.ops import MLPArch, BatchedMLP
from quinn_amd.mcmc.device_amcmc import DeviceAMCMC

C, N = 64, 4096
NMCMC = int(os.environ.get("NMCMC", "5000"))
arch = MLPArch((1, 64, 64, 64, 1), "tanh")
rs = np.random.RandomState(0)
x = rs.rand(N, 1) * 2 * np.pi - np.pi
y = 0.02 * rs.randn(N, 1) + np.sin(x)
op = BatchedMLP(arch, x, y)
ini = np.stack([np.random.RandomState(1000 + c).rand(arch.nparams) for c in range(C)])
UG = os.environ.get("USE_GRAPH", "0") == "1"
res = {"use_graph": UG, "nmcmc": NMCMC}
FUSE = os.environ.get("QN_FUSE", "1") == "1"
res["fuse_propose"] = FUSE
MAXROWS = int(os.environ.get("MAX_ROWS", "4096"))       # bound on the stored history rows per chain (compressed beyond)
STORE = os.environ.get("STORE_CHAIN", "1" if NMCMC <= 12000 else "0") == "1"     # 64 x 50001 x 8513 doubles do not fit
res["max_rows"], res["store_chain"] = MAXROWS, STORE
GROUPS = int(os.environ.get("NGROUPS", "0")) or None   # chain groups on their own HIP streams (default: the engine's, 2 at 64 chains)
eng = DeviceAMCMC(op, 0.02, gamma=0.01, t0=100, tadapt=1000, seed=1, use_graph=UG, fuse_propose=FUSE, max_rows=MAXROWS,
                  groups=GROUPS)
eng.run(20, ini, store_chain=True)                      # warm-up (first launches)
# warm the caching allocator with the run's two large buffers (chain f64, state history f32): a fresh
# hipMalloc of ~33 GB costs several hundred ms and is not part of the stepping rate
_a = torch.empty(C, NMCMC + 1, arch.nparams, dtype=torch.float64, device=op.device) if STORE else None
_b = torch.empty(C, min(NMCMC + 1, MAXROWS), (arch.nparams + 3) // 4 * 4, dtype=torch.float16, device=op.device)
del _a, _b
res["groups"] = NG = eng._ngroups(C)
for e in (eng._subs[0] if NG > 1 else [eng]):          # adapted-phase buffers + first use of the solver paths (set-up, untimed)
    e.prepare(NMCMC, C // NG)
marks = []


class Tick:
    """wall-clock at every 1000 steps (the engine prints there when verbose)"""
    def write(self, s):
        if "completed" in s:
            # (the engine prints from the thread that enqueues chain group 0, inside that group's stream context: wait for THAT
            # stream only -- a device-wide wait would also wait for whatever the other group's thread keeps enqueuing)
            torch.cuda.current_stream().synchronize(); marks.append(time.perf_counter())
    def flush(self):
        pass


torch.cuda.synchronize(); t0 = time.perf_counter()
old = sys.stdout; sys.stdout = Tick()
try:
    r = eng.run(NMCMC, ini, store_chain=STORE, verbose=True)
finally:
    sys.stdout = old
torch.cuda.synchronize(); tt = time.perf_counter() - t0
marks = [t0] + marks
per = [1000.0 / (b - a) for a, b in zip(marks[:-1], marks[1:])]
res["steps_per_s_by_1000"] = [round(v, 1) for v in per]
res["phaseA_steps_per_s"] = per[0]
res["phaseA_logpost_evals_per_s"] = per[0] * C
res["adapted_steps_per_s_last_window"] = per[-1]
res["adapted_logpost_evals_per_s_last_window"] = per[-1] * C
res["overall_steps_per_s"] = NMCMC / tt
res["overall_logpost_evals_per_s"] = NMCMC * C / tt
acc = r["accrate"]
res["accrate_mean"] = float(acc.mean()); res["accrate_min"] = float(acc.min()); res["accrate_max"] = float(acc.max())
lp = r["logpost"]
res["logpost_start_mean"] = float(lp[:, 0].mean()); res["logpost_end_mean"] = float(lp[:, -1].mean())
res["peak_mem_GB"] = torch.cuda.max_memory_allocated() / 1e9
res["history_GB"] = C * min(NMCMC + 1, MAXROWS) * ((arch.nparams + 3) // 4 * 4) * 2 / 1e9
_ls = eng.last_state or eng.last_states[0]
res["rows_in_use_end"] = [int(v) + 1 for v in _ls['kcur'][_ls['par']].cpu().numpy()[:8]]
print(json.dumps(res))
