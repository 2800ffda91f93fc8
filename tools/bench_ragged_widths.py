import sys, os, time, json, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from quinn_amd import _lib
from quinn_amd.ops import MLPArch, BatchedMLP
def timeit(fn, n):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
rs = np.random.RandomState(0)
N, B = 4096, 64
DT = os.environ.get("QN_DTYPE", "float64")
out = {}
for dims in [(1, 11, 11, 11, 1), (1, 50, 50, 50, 1), (1, 20, 40, 10, 1), (1, 64, 64, 64, 1),
             (1, 100, 100, 100, 1), (1, 200, 200, 200, 1), (1, 500, 500, 1),    # > 64: padded to 128 / multiples of 64
             (6, 50, 50, 50, 7), (6, 33, 33, 7)]:                                # fused kernels do not apply: layer-wise on the twin
    arch = MLPArch(dims, "tanh")
    x = rs.rand(N, dims[0]) * 6 - 3; y = np.sin(x).sum(axis=1, keepdims=True) * np.ones((1, dims[-1]))
    op = BatchedMLP(arch, x, y, dtype=DT)
    W = op.weights(0.1 * rs.randn(B, arch.nparams))
    r = {}
    for name, path in (("generic", _lib.PATH_GENERIC), ("auto", _lib.PATH_AUTO)):
        old = op.set_path(path)
        try:
            r[name + "_fwd_evals_per_s"] = B / timeit(lambda: op.sse(W), 30)
            r[name + "_grad_evals_per_s"] = B / timeit(lambda: op.sse_grad(W), 20)
            r[name + "_path"] = op.path(B, N, False)
        finally:
            op.set_path(old)
    out[str(dims)] = r
print(json.dumps(out, indent=1))
