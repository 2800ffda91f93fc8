"""Deep saturated networks (weights ~ N(0, 16), five 128-wide layers) amplify rounding: gradient of the int8-slice kernels and
of the float64 kernels against the oracle (numpy float64), and against each other.  See DESIGN.md section 4.2a (accuracy contract)."""
import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from oracle import mlp_ref
from quinn_amd import _lib
from quinn_amd.ops import MLPArch, BatchedMLP
for seed in range(4):
    rs = np.random.RandomState(seed)
    dims = (4, 128, 128, 128, 128, 128, 1); N = 8; B = 2
    arch = MLPArch(dims, "tanh", bias=True)
    x = rs.rand(N, 4) * 2 * np.pi - np.pi; y = np.sin(x).sum(axis=1, keepdims=True)
    W = 4.0 * rs.randn(B, arch.nparams)
    op = BatchedMLP(arch, x, y)
    res = {}
    for path in (_lib.PATH_AUTO, _lib.PATH_GENERIC):
        op.set_path(path); s, g = op.sse_grad(W); res[path] = (s.cpu().numpy(), g.cpu().numpy())
    mod = mlp_ref.build_module(mlp_ref.MLPSpec(dims, "tanh"))
    for b in range(B):
        gref = mlp_ref.logpostgrad(mod, W[b], x, [v for v in y], 1.0)       # = -0.5 * dSSE/dw
        for name, path in (("int8-slice", _lib.PATH_AUTO), ("float64 kernels", _lib.PATH_GENERIC)):
            gg = -0.5 * res[path][1][b]
            print(seed, b, name, "vs oracle: %.2e" % (np.abs(gg - gref).max() / np.abs(gref).max()), end="   ")
        print("int8 vs f64 kernels: %.2e" % (np.abs(res[_lib.PATH_AUTO][1][b] - res[_lib.PATH_GENERIC][1][b]).max() / np.abs(res[_lib.PATH_GENERIC][1][b]).max()))
