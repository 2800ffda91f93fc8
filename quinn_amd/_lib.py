"""ctypes binding of the C-ABI library (include/quinn_amd.h) and its in-tree build.

The product path has no CPU fallback: if libquinn_amd.so is missing or a call fails,
an exception is raised.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIBDIR = os.path.join(_HERE, "lib")
LIBPATH = os.path.join(LIBDIR, "libquinn_amd.so")
SOURCES = ["qn_api.hip", "qn_generic.hip", "qn_fused.hip", "qn_fused_d8.hip", "qn_fused_o16.hip", "qn_fused_i8.hip", "qn_fused_bwd_i8.hip", "qn_wide_i8.hip", "qn_wide_u_i8.hip", "qn_dw_i8.hip", "qn_mcmc.hip", "qn_rnet.hip"]
# sources that #include another source (the second object of a file compiled in two parts): rebuilt when that one changes
SOURCE_DEPS = {"qn_wide_u_i8.hip": ["qn_wide_i8.hip"], "qn_fused_d8.hip": ["qn_fused.hip"], "qn_fused_o16.hip": ["qn_fused.hip"]}

QN_F64, QN_F32 = 0, 1
ACT_CODES = {"identity": 0, "tanh": 1, "relu": 2}
PATH_AUTO, PATH_GENERIC, PATH_FUSED, PATH_FUSED_DP = 0, 1, 2, 3
ARITH_PLAIN, ARITH_I8_FUSED, ARITH_I8_WIDE, ARITH_I8_LAYERS = 0, 1, 2, 3

# every symbol include/quinn_amd.h declares (tests check the .so exports all of them)
SYMBOLS = ["qn_mlp_desc_create", "qn_rnet_desc_create", "qn_rnet_desc_set_uses", "qn_mlp_desc_destroy", "qn_mlp_num_params", "qn_workspace_bytes",
           "qn_mlp_path", "qn_mlp_arith", "qn_mlp_desc_set_path", "qn_mlp_desc_set_plan_batch", "qn_mlp_sse_fwd", "qn_mlp_sse_parts", "qn_mlp_sse_fwd_parts", "qn_mlp_sse_fwdbwd", "qn_vi_sample_kl",
           "qn_vi_grad", "qn_adam_batched", "qn_mcmc_propose", "qn_mcmc_propose_hist", "qn_mcmc_hist_block_steps", "qn_mcmc_hist_block_coef_bytes", "qn_mcmc_propose_hist_block",
           "qn_mcmc_apply_delta", "qn_mcmc_accept", "qn_mcmc_accept_propose", "qn_hmc_parts", "qn_hmc_begin", "qn_hmc_leap", "qn_hmc_accept", "qn_pred_moments", "qn_debug_tanh", "qn_debug_tanh_finite", "qn_debug_tanh_table", "qn_last_error",
           "qn_version"]


class QuinnAmdError(RuntimeError):
    pass


def build(force=False, verbose=False, jobs=None):
    """Compile the HIP sources for gfx950 into quinn_amd/lib/libquinn_amd.so (hipcc
    cross-compiles without a GPU).  One object per source under quinn_amd/lib/obj/, compiled in
    parallel and only when the source or a header is newer than its object; then one link.
    With QN_HIPCC_FLAGS set (A/B builds, -DQN_...) objects and library go to their OWN directory / file name, keyed by the
    flags (quinn_amd/lib/obj_<key>/, libquinn_amd_<key>.so; select the result with QUINN_AMD_LIB): a flagged build never
    replaces or poisons the default library.  Concurrent callers (ranks, pytest-xdist) are serialised by a file lock."""
    import fcntl
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    hdrs = [os.path.join(CSRC, h) for h in sorted(os.listdir(CSRC)) if h.endswith(".h")] + \
           [os.path.join(_HERE, "..", "include", "quinn_amd.h")]
    hnew = max(os.path.getmtime(h) for h in hdrs)
    extra = os.environ.get("QN_HIPCC_FLAGS", "").split()          # A/B builds (-DQN_...)
    key = hashlib.sha1(" ".join(extra).encode()).hexdigest()[:10] if extra else ""
    libpath = os.path.join(LIBDIR, f"libquinn_amd_{key}.so") if extra else LIBPATH
    objdir = os.path.join(LIBDIR, f"obj_{key}" if extra else "obj")
    os.makedirs(objdir, exist_ok=True)
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and os.path.exists(libpath):
            if os.path.getmtime(libpath) >= max(hnew, max(os.path.getmtime(s) for s in srcs)):
                return libpath
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        # -amdgpu-mfma-vgpr-form: keep MFMA accumulators in VGPRs (gfx950 has one unified register file);
        # without it hipcc copies every loop-carried accumulator VGPR<->AGPR per iteration (25 % of the GEMM loop)
        flags = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form", "-fPIC"]

        def one(src):
            obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
            newest = max([hnew, os.path.getmtime(src)] +
                         [os.path.getmtime(os.path.join(CSRC, dep)) for dep in SOURCE_DEPS.get(os.path.basename(src), [])])
            if not force and os.path.exists(obj) and os.path.getmtime(obj) >= newest:
                return obj
            cmd = [hipcc] + flags + extra + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
            return obj

        with ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as ex:
            objs = list(ex.map(one, srcs))
        cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", libpath] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return libpath


_lib = None


def lib():
    """The loaded library (loads on first use; raises QuinnAmdError if it is not built)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("QUINN_AMD_LIB", LIBPATH)      # A/B builds of the kernels (tools/ab_build.sh)
    if not os.path.exists(path):
        raise QuinnAmdError(f"{path} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    # torch ships its own HIP runtime (torch/lib/libamdhip64.so, same SONAME as /opt/rocm's).
    # It must be the one already mapped when this library is loaded, otherwise the process ends
    # up with two runtimes and ours sees no device / cannot share torch's streams and memory.
    import torch  # noqa: F401
    L = ctypes.CDLL(path)
    vp, i32, i64, f64, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double, ctypes.c_size_t
    L.qn_mlp_desc_create.argtypes = [ctypes.POINTER(i32), i32, i32, i32, ctypes.POINTER(vp)]
    L.qn_mlp_desc_create.restype = i32
    L.qn_rnet_desc_create.argtypes = [i32, i32, i32, i32, i32, ctypes.POINTER(ctypes.c_double), i32, i32, i32, i32,
                                      i32, ctypes.POINTER(vp)]
    L.qn_rnet_desc_create.restype = i32
    L.qn_rnet_desc_set_uses.argtypes = [vp, ctypes.POINTER(ctypes.c_ubyte), i32]
    L.qn_rnet_desc_set_uses.restype = i32
    L.qn_mlp_desc_destroy.argtypes = [vp]
    L.qn_mlp_desc_destroy.restype = i32
    L.qn_mlp_num_params.argtypes = [vp]
    L.qn_mlp_num_params.restype = i64
    L.qn_workspace_bytes.argtypes = [vp, i32, i32, i32, i32]
    L.qn_workspace_bytes.restype = sz
    L.qn_mlp_path.argtypes = [vp, i32, i32, i32, i32]
    L.qn_mlp_path.restype = i32
    L.qn_mlp_desc_set_plan_batch.argtypes = [vp, i32]
    L.qn_mlp_desc_set_plan_batch.restype = i32
    L.qn_mlp_arith.argtypes = [vp, i32, i32, i32, i32]
    L.qn_mlp_arith.restype = i32
    L.qn_mlp_desc_set_path.argtypes = [vp, i32]
    L.qn_mlp_desc_set_path.restype = i32
    L.qn_mlp_sse_fwd.argtypes = [vp, i32, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, sz, vp]
    L.qn_mlp_sse_fwd.restype = i32
    L.qn_mlp_sse_parts.argtypes = [vp, i32, i32, i32]
    L.qn_mlp_sse_parts.restype = i32
    L.qn_mlp_sse_fwd_parts.argtypes = [vp, i32, vp, vp, vp, vp, i32, i32, i32, vp, vp, sz, vp]
    L.qn_mlp_sse_fwd_parts.restype = i32
    L.qn_mlp_sse_fwdbwd.argtypes = [vp, i32, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, sz, vp]
    L.qn_mlp_sse_fwdbwd.restype = i32
    L.qn_vi_sample_kl.argtypes = [vp, vp, vp, i32, i64, f64, f64, f64, i32, vp, vp, vp, vp]
    L.qn_vi_sample_kl.restype = i32
    L.qn_vi_grad.argtypes = [vp, vp, vp, vp, i32, i64, f64, f64, f64, f64, f64, i32, vp, vp, vp]
    L.qn_vi_grad.restype = i32
    L.qn_adam_batched.argtypes = [vp, vp, vp, vp, vp, i32, i64, i32, f64, f64, f64, f64, f64, i32, vp]
    L.qn_adam_batched.restype = i32
    u64 = ctypes.c_uint64
    L.qn_mcmc_propose.argtypes = [vp, vp, f64, i32, i32, i64, u64, vp, vp, vp]
    L.qn_mcmc_propose.restype = i32
    L.qn_mcmc_propose_hist.argtypes = [vp, vp, vp, vp, vp, vp, f64, f64, i32, i32, i64, i64, i32, u64, vp, vp, vp]
    L.qn_mcmc_propose_hist.restype = i32
    L.qn_mcmc_hist_block_steps.argtypes = []
    L.qn_mcmc_hist_block_steps.restype = i32
    L.qn_mcmc_hist_block_coef_bytes.argtypes = [i32, i32]
    L.qn_mcmc_hist_block_coef_bytes.restype = sz
    L.qn_mcmc_propose_hist_block.argtypes = [vp, vp, vp, vp, vp, f64, f64, i32, i32, i64, i64, i32, u64, i64, vp, vp, vp, vp, vp]
    L.qn_mcmc_propose_hist_block.restype = i32
    L.qn_mcmc_apply_delta.argtypes = [vp, vp, i32, f64, i32, i32, i64, u64, vp, vp, vp]
    L.qn_mcmc_apply_delta.restype = i32
    L.qn_mcmc_accept.argtypes = [vp, vp, f64, i32, i32, i32, i64, i32, u64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                 vp, i32, i64, vp, i32, i32, vp]
    L.qn_mcmc_accept.restype = i32
    L.qn_mcmc_accept_propose.argtypes = [vp, vp, f64, i32, i32, i32, i64, i32, u64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                         vp, vp, vp, i32, i64, vp, i32, vp, f64, vp, i32, f64, vp, i32, i32, vp]
    L.qn_mcmc_accept_propose.restype = i32
    L.qn_hmc_parts.argtypes = [i64]
    L.qn_hmc_parts.restype = i32
    L.qn_hmc_begin.argtypes = [vp, vp, f64, f64, i32, i32, i64, u64, vp, vp, vp, vp, vp]
    L.qn_hmc_begin.restype = i32
    L.qn_hmc_leap.argtypes = [vp, i32, f64, f64, i32, i32, i64, vp, vp, vp, vp]
    L.qn_hmc_leap.restype = i32
    L.qn_hmc_accept.argtypes = [vp, vp, vp, vp, vp, f64, i32, i32, i32, i64, i32, u64, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                vp, i32, vp]
    L.qn_hmc_accept.restype = i32
    L.qn_pred_moments.argtypes = [vp, i32, i64, i64, vp, vp, vp]
    L.qn_pred_moments.restype = i32
    L.qn_debug_tanh.argtypes = [vp, vp, i64, vp]
    L.qn_debug_tanh.restype = i32
    L.qn_debug_tanh_finite.argtypes = [vp, vp, i64, vp]
    L.qn_debug_tanh_finite.restype = i32
    L.qn_debug_tanh_table.argtypes = [vp, vp, i64, i32, vp]
    L.qn_debug_tanh_table.restype = i32
    L.qn_last_error.restype = ctypes.c_char_p
    L.qn_version.restype = ctypes.c_char_p
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        raise QuinnAmdError(f"{what} failed (code {rc}): {lib().qn_last_error().decode()}")
