#!/usr/bin/env python3
"""One-GPU predictor of `bench.py --scaling strong`: the cfg2 log-posterior launch (3x64 tanh, N = 4096) with B = 64 / 32 / 16 / 8
chains -- what a rank sees at 1 / 2 / 4 / 8 GPUs when 64 chains are block-partitioned -- timed as HIP-graph replays at a settled
clock (the step itself replayed for 250 ms first).  Prints ms per launch, the ratio to B = 64 and the strong-scaling
efficiency (B / 64) / ratio.  usage: tools/bench_strong.py [--kind logpost|grad]"""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quinn_amd.ops import MLPArch, BatchedMLP

kind = sys.argv[sys.argv.index("--kind") + 1] if "--kind" in sys.argv else "logpost"
arch = MLPArch((1, 64, 64, 64, 1), "tanh")
rs = np.random.RandomState(0)
x = rs.rand(4096, 1) * 2 * np.pi - np.pi
y = np.sin(x) + 0.02 * rs.randn(4096, 1)
op = BatchedMLP(arch, x, y)
dev = op.device
res = {}
for B in (64, 32, 16, 8, 64):
    Ws = [op.weights(0.1 * np.random.RandomState(100 + k).randn(B, arch.nparams)) for k in range(4)]
    fn = (lambda W: op.sse(W)) if kind == "logpost" else (lambda W: op.sse_grad(W))
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for W in Ws: fn(W)
    torch.cuda.current_stream(dev).wait_stream(side)
    g = torch.cuda.CUDAGraph()
    NG = 40
    with torch.cuda.graph(g):
        for i in range(NG): r = fn(Ws[i % 4])
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.25:
        g.replay(); torch.cuda.synchronize(dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in ev:
        a.record(); g.replay(); b.record()
    torch.cuda.synchronize(dev)
    ms = np.median([a.elapsed_time(b) for a, b in ev]) / NG
    res.setdefault(B, []).append(ms)
    del g
base = min(res[64])
print(f"kind {kind}: ms per launch (HIP-graph replays, median of 20 x 40 launches)")
for B in (64, 32, 16, 8):
    ms = min(res[B])
    print(f"  B = {B:2d}: {ms:.4f} ms   ratio to B = 64: {ms / base:.3f}   strong-scaling efficiency at {64 // B} GPU(s): {100 * (B / 64) / (ms / base):.0f} %", flush=True)
print(json.dumps({"kind": kind, "ms": {str(k): min(v) for k, v in res.items()}}))
