#!/usr/bin/env python3
"""Throughput of the batched operator at the shapes of BASELINE.json configs 1..5 (reduced chain
counts where the full batch would not fit one pass): evals/s and algorithmic TFLOP/s, float64."""
import sys, os, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quinn_amd.ops import MLPArch, BatchedMLP
CFG = {  # name: dims, N, B
    "cfg1 2x16 N=256 B=64": ((1, 16, 16, 1), 256, 64),
    "cfg2 3x64 N=4096 B=64": ((1, 64, 64, 64, 1), 4096, 64),
    "cfg3 3x128 N=8192 S=128": ((2, 128, 128, 128, 1), 8192, 128),
    "cfg4 4x256 N=16384 M=64": ((1, 256, 256, 256, 256, 1), 16384, 64),
    "cfg5 4x256 N=32768 C=32": ((1, 256, 256, 256, 256, 1), 32768, 32),
}
def timeit(fn, n):
    # the GPU clock of a fresh process ramps up over the first ~50 ms of load (profiles/r03_clock_ramp_kernel_trace.txt):
    # run the call for 0.3 s first, then time n calls (at least 0.2 s worth)
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    n = max(n, int(0.2 / max(time.perf_counter() - t0, 1e-6)))
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n
out = {}
DT = "float32" if len(sys.argv) > 1 and sys.argv[1] == "f32" else "float64"
for name, (dims, N, B) in CFG.items():
    arch = MLPArch(dims, "tanh")
    rs = np.random.RandomState(0)
    x = rs.rand(N, dims[0]) * 6 - 3; y = np.sin(x).sum(axis=1, keepdims=True)
    op = BatchedMLP(arch, x, y, dtype=DT)
    W = op.weights(0.1 * rs.randn(B, arch.nparams))
    tf = timeit(lambda: op.sse(W), 5)
    tg = timeit(lambda: op.sse_grad(W), 3)
    out[name] = {"path_fwd": op.path(B), "path_grad": op.path(B, want_grad=True),
                 "fwd_evals_per_s": B / tf, "fwd_tflops": B * arch.flops_fwd(N) / tf / 1e12,
                 "grad_evals_per_s": B / tg, "grad_tflops": B * arch.flops_fwdbwd(N) / tg / 1e12}
    print(name, json.dumps(out[name]), flush=True)
