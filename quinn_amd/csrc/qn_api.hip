// C-ABI entry points (include/quinn_amd.h): descriptor, dispatch between the kernel
// families, and the small elementwise kernels of the VI / ensemble trainers.
#include "qn_common.h"
#include "qn_fused_args.h"
#include "qn_math.h"
#include <cmath>
#include <cstring>

static thread_local char g_err[512] = "";

void qn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* qn_last_error(void) { return g_err; }
extern "C" const char* qn_version(void) { return "quinn_amd 0.1 gfx950"; }
extern "C" int qn_mlp_desc_set_path(qn_desc* d, int path) {
    if (!d) return QN_EINVAL;
    const int old = d->path;
    if (path == QN_PATH_AUTO || path == QN_PATH_GENERIC || path == QN_PATH_FUSED || path == QN_PATH_FUSED_DP) {
        d->path = path;
        if (d->padded) d->padded->path = path;         // the fused kernels run on the zero-padded twin
    }
    return old;
}

extern "C" int qn_mlp_desc_set_plan_batch(qn_desc* d, int batch) {
    if (!d) return QN_EINVAL;
    const int old = d->plan_batch;
    if (batch >= 0) {
        d->plan_batch = batch;
        if (d->padded) d->padded->plan_batch = batch;
    }
    return old;
}

extern "C" int qn_mlp_desc_create(const int* dims, int ndims, int act, int has_bias, qn_desc** out) {
    if (!dims || !out || ndims < 2 || ndims > QN_MAX_LAYERS + 1) {
        qn_set_error("qn_mlp_desc_create: need 2 <= ndims <= %d", QN_MAX_LAYERS + 1);
        return QN_EINVAL;
    }
    if (act != QN_ACT_IDENTITY && act != QN_ACT_TANH && act != QN_ACT_RELU) {
        qn_set_error("qn_mlp_desc_create: unknown activation %d", act);
        return QN_EINVAL;
    }
    qn_desc* d = new qn_desc();
    d->nlayers = ndims - 1;
    d->act = act;
    d->has_bias = has_bias ? 1 : 0;
    d->hmax = 0;
    int64_t off = 0;
    for (int i = 0; i < ndims; ++i) {
        if (dims[i] <= 0) {
            delete d;
            qn_set_error("qn_mlp_desc_create: dims[%d] = %d", i, dims[i]);
            return QN_EINVAL;
        }
        d->dims[i] = dims[i];
        if (i > 0 && dims[i] > d->hmax) d->hmax = dims[i];
    }
    for (int l = 0; l < d->nlayers; ++l) {
        d->offW[l] = off;
        off += (int64_t)d->dims[l] * d->dims[l + 1];
        d->offB[l] = off;
        if (d->has_bias) off += d->dims[l + 1];
    }
    d->p = off;
    d->kind = QN_KIND_MLP;
    d->padded = nullptr;
    // zero-padded twin of the network (hidden units only; padded units stay exactly 0):
    //   widths <= 64 that are not one common 16 / 32 / 64   -> that common width, for the fused kernels
    //   widths <= 128 with one that is no multiple of 64     -> 128 everywhere (streaming fused forward, MFMA GEMMs)
    //   wider, with a width that is no multiple of 64        -> every width rounded up to a multiple of 64, so that
    //                                                           the layer-wise path runs its MFMA GEMMs, not the VALU kernels
    if (d->nlayers >= 2) {
        int hmax = 0;
        bool uniform = true, mult64 = true;
        for (int l = 1; l < d->nlayers; ++l) {
            hmax = d->dims[l] > hmax ? d->dims[l] : hmax;
            uniform = uniform && d->dims[l] == d->dims[1];
            mult64 = mult64 && d->dims[l] % 64 == 0;
        }
        const int H = hmax <= 16 ? 16 : hmax <= 32 ? 32 : hmax <= 64 ? 64 : 128;
        if (hmax <= 64 ? !(uniform && hmax == H) : !mult64) {
            qn_desc* q = new qn_desc(*d);
            int64_t o2 = 0;
            for (int l = 1; l < q->nlayers; ++l) q->dims[l] = hmax <= 128 ? H : (d->dims[l] + 63) / 64 * 64;
            for (int l = 0; l < q->nlayers; ++l) {
                q->offW[l] = o2;
                o2 += (int64_t)q->dims[l] * q->dims[l + 1];
                q->offB[l] = o2;
                if (q->has_bias) o2 += q->dims[l + 1];
            }
            q->p = o2;
            q->hmax = 0;
            for (int l = 1; l <= q->nlayers; ++l) q->hmax = q->dims[l] > q->hmax ? q->dims[l] : q->hmax;
            q->padded = nullptr;
            d->padded = q;
        }
    }
    *out = d;
    return QN_OK;
}

extern "C" int qn_rnet_desc_create(int indim, int rdim, int outdim, int nsteps, int npar, const double* coef,
                                   int act, int has_bias, int layer_pre, int layer_post, int mlp, qn_desc** out) {
    if (!coef || !out || indim <= 0 || rdim <= 0 || outdim <= 0 || nsteps <= 0 || npar <= 0 ||
        nsteps > QN_MAX_LAYERS || npar > QN_MAX_LAYERS) {
        qn_set_error("qn_rnet_desc_create: need positive sizes, nsteps <= %d, npar <= %d", QN_MAX_LAYERS,
                     QN_MAX_LAYERS);
        return QN_EINVAL;
    }
    if (act != QN_ACT_IDENTITY && act != QN_ACT_TANH) {
        qn_set_error("qn_rnet_desc_create: activation %d (tanh or identity, rnet.py:123-126)", act);
        return QN_EINVAL;
    }
    if ((indim != rdim && !layer_pre) || (outdim != rdim && !layer_post)) {   // rnet.py:85-88
        qn_set_error("qn_rnet_desc_create: indim/outdim != rdim needs layer_pre/layer_post");
        return QN_EINVAL;
    }
    qn_desc* d = new qn_desc();
    d->kind = QN_KIND_RNET;
    d->nlayers = 2;
    d->dims[0] = indim; d->dims[1] = rdim; d->dims[2] = outdim;
    d->hmax = rdim > outdim ? rdim : outdim;
    d->act = act;
    d->has_bias = has_bias ? 1 : 0;
    d->rn_r = rdim; d->rn_steps = nsteps; d->rn_npar = npar;
    d->rn_pre = layer_pre ? 1 : 0; d->rn_post = layer_post ? 1 : 0; d->rn_mlp = mlp ? 1 : 0;
    for (int i = 0; i < nsteps * npar; ++i) d->rn_coef[i] = coef[i];
    for (int i = 0; i < QN_MAX_LAYERS * QN_MAX_LAYERS; ++i) d->rn_uses[i] = 1;
    // parameters() order of the module: weight_pre, bias_pre, weight_post, bias_post, ww_*, bb_*  (rnet.py:90-120)
    int64_t off = 0;
    d->rn_offWpre = off; if (layer_pre) off += (int64_t)rdim * indim;
    d->rn_offBpre = off; if (layer_pre) off += rdim;
    d->rn_offWpost = off; if (layer_post) off += (int64_t)outdim * rdim;
    d->rn_offBpost = off; if (layer_post) off += outdim;
    d->rn_offWW = off; off += (int64_t)npar * rdim * rdim;
    d->rn_offBB = off; if (has_bias) off += (int64_t)npar * rdim;
    d->p = off;
    d->padded = nullptr;
    *out = d;
    return QN_OK;
}

extern "C" int qn_rnet_desc_set_uses(qn_desc* d, const unsigned char* uses, int n) {
    if (!d || !uses || d->kind != QN_KIND_RNET || n != d->rn_steps * d->rn_npar) {
        qn_set_error("qn_rnet_desc_set_uses: need a residual-network descriptor and nsteps * npar entries");
        return QN_EINVAL;
    }
    for (int i = 0; i < n; ++i) {
        if (!uses[i] && d->rn_coef[i] != 0.0) {
            qn_set_error("qn_rnet_desc_set_uses: entry %d is marked unused but has coefficient %g", i, d->rn_coef[i]);
            return QN_EINVAL;
        }
        d->rn_uses[i] = uses[i] ? 1 : 0;
    }
    return QN_OK;
}

extern "C" int qn_mlp_desc_destroy(qn_desc* d) {
    if (d) delete d->padded;
    delete d;
    return QN_OK;
}

extern "C" int64_t qn_mlp_num_params(const qn_desc* d) { return d ? d->p : -1; }

// the descriptor the fused kernels would run: the network itself, or its zero-padded twin
static const qn_desc* fused_desc(const qn_desc* d) { return d->padded ? d->padded : d; }
static bool fused_ok(const qn_desc* d, int B, int Nb, int want_grad, int dtype) {
    return d->kind == QN_KIND_MLP && qn_fused_supported(fused_desc(d), B, Nb, want_grad, dtype);
}
// 128-wide float64 tanh networks, forward only: the fused int8-slice forward of the layer-wise family (qn_wide_i8.hip,
// one launch) is faster than the float64-MFMA streaming kernel (cfg3 shape: 1.25 vs 1.52 ms); QN_PATH_FUSED keeps the latter
static bool prefer_wide(const qn_desc* d, int want_grad, int dtype) {
    return d->path == QN_PATH_AUTO && !want_grad && dtype == QN_F64 && d->kind == QN_KIND_MLP &&
           qn_i8_wide_applies(d->padded ? d->padded : d);        // (a zero-padded twin runs through run_padded)
}
static bool use_fused(const qn_desc* d, int B, int Nb, int want_grad, int dtype) {
    if (d->path == QN_PATH_GENERIC || prefer_wide(d, want_grad, dtype)) return false;
    return fused_ok(d, B, Nb, want_grad, dtype);
}
// head of the workspace of a run on the zero-padded twin: padded weights (+ padded gradient) | flags [B + 1][64] | scratch of the
// exceptional-value fix-up (k_padded_fixup below): per chain every layer's outputs + two dz vectors
static size_t padded_fixup_doubles(const qn_desc* d) {
    size_t sum = 0, hmax = 0;
    for (int l = 1; l <= d->nlayers; ++l) {
        sum += (size_t)d->dims[l];
        hmax = (size_t)d->dims[l] > hmax ? (size_t)d->dims[l] : hmax;
    }
    return sum + 2 * hmax;
}
static size_t padded_head_bytes(const qn_desc* d, int B, int want_grad, size_t el) {
    return (want_grad ? 2 : 1) * qn_align((size_t)B * d->padded->p * el) + qn_align((size_t)(B + 1) * 64 * sizeof(int)) +
           qn_align((size_t)B * padded_fixup_doubles(d) * el);
}
static size_t fused_ws(const qn_desc* d, int B, int Nb, int want_grad, int dtype) {
    size_t tot = qn_fused_workspace(fused_desc(d), B, Nb, want_grad, dtype);
    if (d->padded) tot += padded_head_bytes(d, B, want_grad, sizeof(double));
    return tot;
}
// layer-wise kernels on the padded twin (hidden widths >= 48 that are no multiples of 64; for widths <= 64 this is
// reached only where the fused kernels do not apply: float32, more than 4 inputs / outputs, deep 64-wide
// gradients): MFMA GEMMs instead of VALU kernels.  Not under a forced path: QN_PATH_GENERIC stays the exact-width
// reference the tests compare against.
static bool use_padded_generic(const qn_desc* d) {
    if (d->kind != QN_KIND_MLP || !d->padded || d->padded->dims[1] < 64 || d->path != QN_PATH_AUTO) return false;
    int hmax = 0;
    for (int l = 1; l < d->nlayers; ++l) hmax = d->dims[l] > hmax ? d->dims[l] : hmax;
    return hmax >= 48;      // measured: 50 -> 64 wins 1.4-2.6x, 40 -> 64 ties, 33 -> 64 loses 30 % against the exact VALU kernels
}
static size_t padded_generic_ws(const qn_desc* d, int B, int Nb, int want_grad, int dtype) {
    const size_t el = dtype == QN_F32 ? sizeof(float) : sizeof(double);
    return qn_generic_workspace(d->padded, B, Nb, want_grad, dtype) + padded_head_bytes(d, B, want_grad, el);
}

// ---- zero-padding of hidden layers (weights in, gradients out); one thread per element, layer found by offset
namespace {
struct PadMap {
    int L, has_bias;
    int hin[QN_MAX_LAYERS], hout[QN_MAX_LAYERS], Hin[QN_MAX_LAYERS], Hout[QN_MAX_LAYERS];
    int64_t off[QN_MAX_LAYERS + 1], offp[QN_MAX_LAYERS + 1], p, pp;
};
// "Padded units stay exactly 0" holds as long as nothing they are multiplied with is Inf / NaN (0 . Inf = NaN): an infinite
// input, or -- relu / identity -- an activation that overflows.  The pad kernel therefore checks every weight of the chain and
// (blockIdx.y == 0) every input against 2^ebound, chosen so that below it no intermediate value can overflow; flagged chains
// (flags[b]; flags[B] = an input: every chain) are recomputed from the ORIGINAL weights by k_padded_fixup after the twin's
// kernels.  Found by tests/fuzz_all.py (x = -inf on a 33-wide network: the reference saturates, the twin returned NaN).
constexpr int PAD_SLOTS = 64;           // blocks per chain of k_pad_weights at most (one flag slot each)
template <typename T> __device__ __forceinline__ bool pad_bounded(T v, T bound) { return (v < T(0) ? -v : v) < bound; }   // (NaN: false)
template <typename T> __global__ void k_pad_weights(PadMap m, const T* __restrict__ W, T* __restrict__ Wp, const T* __restrict__ X,
                                                    int64_t nx, const T* __restrict__ Y, int64_t ny, T bound, int* __restrict__ flags) {
    // every block WRITES its own slot flags[(chain or B) * PAD_SLOTS + blockIdx.x] (no atomics, no memset node ahead of the kernel)
    const int b = blockIdx.y;
    int bad = 0;
    if (b == 0) {
        int xb = 0;
        for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nx; e += (int64_t)gridDim.x * blockDim.x) xb |= !pad_bounded(X[e], bound);
        // (the targets too: an infinite target makes the last layer's dz infinite, and 0 . Inf = NaN in the padded columns of the
        // backward pass where the reference's gradient is +-Inf)
        for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < ny; e += (int64_t)gridDim.x * blockDim.x) xb |= !pad_bounded(Y[e], bound);
        xb = __syncthreads_or(xb);
        if (threadIdx.x == 0) flags[(int64_t)gridDim.y * PAD_SLOTS + blockIdx.x] = xb;
    }
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < m.p; e += (int64_t)gridDim.x * blockDim.x)
        bad |= !pad_bounded(W[(int64_t)b * m.p + e], bound);
    bad = __syncthreads_or(bad);
    if (threadIdx.x == 0) flags[(int64_t)b * PAD_SLOTS + blockIdx.x] = bad;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < m.pp; e += (int64_t)gridDim.x * blockDim.x) {
        int l = 0;
        while (l + 1 < m.L && e >= m.offp[l + 1]) ++l;
        const int64_t loc = e - m.offp[l], nW = (int64_t)m.Hout[l] * m.Hin[l];
        T v = 0;
        if (loc < nW) {
            const int j = (int)(loc / m.Hin[l]), i = (int)(loc % m.Hin[l]);
            if (j < m.hout[l] && i < m.hin[l]) v = W[(int64_t)b * m.p + m.off[l] + (int64_t)j * m.hin[l] + i];
        } else {
            const int j = (int)(loc - nW);
            if (j < m.hout[l]) v = W[(int64_t)b * m.p + m.off[l] + (int64_t)m.hout[l] * m.hin[l] + j];
        }
        Wp[(int64_t)b * m.pp + e] = v;
    }
}
struct FixNet {
    int L, has_bias, act, d, o;
    int dims[QN_MAX_LAYERS + 1];
    int64_t offW[QN_MAX_LAYERS], offB[QN_MAX_LAYERS], p;
    int acto[QN_MAX_LAYERS], dzo, hmax, per_chain;        // offsets into a chain's scratch
};
__device__ __forceinline__ double fix_tanh(double z) { return qn_tanh_f64(z); }
__device__ __forceinline__ float fix_tanh(float z) { return qn_tanh_f32(z); }
// Flagged chains again, from the original (unpadded) weights, one data row at a time with the whole workgroup: forward,
// residual, backward (thread = its own gradient entries: no atomics, a fixed summation order).  The plain loops of the
// reference's ops (F.linear, activation, autograd); slow, and only ever run for chains with unbounded values.
template <typename T>
__global__ __launch_bounds__(256) void k_padded_fixup(FixNet s, const T* __restrict__ W, const T* __restrict__ X, const T* __restrict__ Y,
                                                      const int32_t* __restrict__ row_idx, int Nb, const int* __restrict__ flags,
                                                      int nslots, double* __restrict__ sse, T* __restrict__ pred, T* __restrict__ grad,
                                                      T* __restrict__ scratch) {
    const int b = blockIdx.x, tid = threadIdx.x;
    int any = 0;
    if (tid < nslots) any = flags[(int64_t)b * PAD_SLOTS + tid] | flags[(int64_t)gridDim.x * PAD_SLOTS + tid];
    if (!__syncthreads_or(any)) return;
    const T* Wb = W + (int64_t)b * s.p;
    T* act = scratch + (int64_t)b * s.per_chain;
    T* dzbuf[2] = {act + s.dzo, act + s.dzo + s.hmax};
    T* G = grad ? grad + (int64_t)b * s.p : nullptr;
    if (G)
        for (int64_t e = tid; e < s.p; e += 256) G[e] = T(0);
    __shared__ double red[256];
    double sacc = 0.0;
    __syncthreads();
    for (int n = 0; n < Nb; ++n) {
        const int64_t rr = row_idx ? (int64_t)row_idx[(int64_t)b * Nb + n] : (int64_t)n;
        for (int l = 0; l < s.L; ++l) {
            const T* in = l == 0 ? X + rr * s.d : act + s.acto[l - 1];
            const int hin = s.dims[l], hout = s.dims[l + 1];
            for (int j = tid; j < hout; j += 256) {
                T z = s.has_bias ? Wb[s.offB[l] + j] : T(0);
                for (int i = 0; i < hin; ++i) z = fma(Wb[s.offW[l] + (int64_t)j * hin + i], in[i], z);
                if (l + 1 < s.L) z = s.act == QN_ACT_TANH ? fix_tanh(z) : (s.act == QN_ACT_RELU ? qn_relu<T>(z) : z);
                act[s.acto[l] + j] = z;
            }
            __syncthreads();
        }
        const T* out = act + s.acto[s.L - 1];
        for (int qo = tid; qo < s.o; qo += 256) {
            const T res = out[qo] - Y[rr * s.o + qo];
            sacc += (double)res * (double)res;
            if (pred) pred[((int64_t)b * Nb + n) * s.o + qo] = out[qo];
            dzbuf[0][qo] = T(2) * res;
        }
        __syncthreads();
        if (G) {
            int cur = 0;
            for (int l = s.L - 1; l >= 0; --l) {
                const T* in = l == 0 ? X + rr * s.d : act + s.acto[l - 1];
                const T* dz = dzbuf[cur];
                const int hin = s.dims[l], hout = s.dims[l + 1];
                for (int e = tid; e < hout * hin; e += 256) G[s.offW[l] + e] = fma(dz[e / hin], in[e % hin], G[s.offW[l] + e]);
                if (s.has_bias)
                    for (int j = tid; j < hout; j += 256) G[s.offB[l] + j] += dz[j];
                if (l > 0)
                    for (int i = tid; i < hin; i += 256) {
                        T a = T(0);
                        for (int j = 0; j < hout; ++j) a = fma(Wb[s.offW[l] + (int64_t)j * hin + i], dz[j], a);
                        dzbuf[cur ^ 1][i] = qn_act_bwd<T>(a, in[i], s.act);
                    }
                __syncthreads();
                cur ^= 1;
            }
        }
    }
    red[tid] = sacc;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int k = 0; k < 256; ++k) t += red[k];
        sse[b] = t;
    }
}
template <typename T> __global__ void k_unpad_grad(PadMap m, const T* __restrict__ Gp, T* __restrict__ G) {
    const int b = blockIdx.y;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < m.p; e += (int64_t)gridDim.x * blockDim.x) {
        int l = 0;
        while (l + 1 < m.L && e >= m.off[l + 1]) ++l;
        const int64_t loc = e - m.off[l], nW = (int64_t)m.hout[l] * m.hin[l];
        int64_t src;
        if (loc < nW) src = m.offp[l] + (loc / m.hin[l]) * m.Hin[l] + loc % m.hin[l];
        else src = m.offp[l] + (int64_t)m.Hout[l] * m.Hin[l] + (loc - nW);
        G[(int64_t)b * m.p + e] = Gp[(int64_t)b * m.pp + src];
    }
}
PadMap pad_map(const qn_desc* d) {
    const qn_desc* q = d->padded;
    PadMap m;
    m.L = d->nlayers; m.has_bias = d->has_bias; m.p = d->p; m.pp = q->p;
    for (int l = 0; l < d->nlayers; ++l) {
        m.hin[l] = d->dims[l]; m.hout[l] = d->dims[l + 1]; m.Hin[l] = q->dims[l]; m.Hout[l] = q->dims[l + 1];
        m.off[l] = d->offW[l]; m.offp[l] = q->offW[l];
    }
    m.off[d->nlayers] = d->p; m.offp[d->nlayers] = q->p;
    return m;
}
// pad -> kernels on the padded twin (fused, or layer-wise when `generic`) -> unpad the gradient
template <typename T>
int run_padded_t(const qn_desc* d, bool generic, int dtype, const T* W, const void* X, const void* Y,
                 const int32_t* row_idx, int B, int N, int Nb, double* sse, void* pred, T* gradW, void* ws,
                 size_t ws_bytes, hipStream_t st) {
    const qn_desc* q = d->padded;
    const size_t wbytes = qn_align((size_t)B * q->p * sizeof(T));
    const size_t head = padded_head_bytes(d, B, gradW != nullptr, sizeof(T));
    if (head > ws_bytes) {
        qn_set_error("workspace too small: need more than %zu bytes, got %zu", head, ws_bytes);
        return QN_EWORKSPACE;
    }
    T* Wp = static_cast<T*>(ws);
    T* Gp = gradW ? reinterpret_cast<T*>(static_cast<char*>(ws) + wbytes) : nullptr;
    int* flags = reinterpret_cast<int*>(static_cast<char*>(ws) + (gradW ? 2 : 1) * wbytes);
    T* fix_scratch = reinterpret_cast<T*>(reinterpret_cast<char*>(flags) + qn_align((size_t)(B + 1) * 64 * sizeof(int)));
    const PadMap m = pad_map(d);
    // 2^e below which no product chain of the network can overflow (tanh: only x . W0; otherwise one factor per layer and up
    // to 2^11 terms per sum)
    const int emax = sizeof(T) == 8 ? 1022 : 126, L = d->nlayers;
    int eb = d->act == QN_ACT_TANH ? emax / 2 - 12 : (emax - 11 * L) / (L + 1);
    eb = eb < 1 ? 1 : eb;
    (void)hipGetLastError();
    int gx = (int)((q->p + 255) / 256);
    const int nslots = gx > PAD_SLOTS ? PAD_SLOTS : gx;
    hipLaunchKernelGGL(k_pad_weights<T>, dim3(nslots, B), dim3(256), 0, st, m, W, Wp, static_cast<const T*>(X),
                       (int64_t)N * d->dims[0], static_cast<const T*>(Y), (int64_t)N * d->dims[L], (T)std::ldexp(1.0, eb), flags);
    const int rc = generic ? qn_generic_run(q, dtype, Wp, X, Y, row_idx, B, N, Nb, sse, pred, Gp,
                                            static_cast<char*>(ws) + head, ws_bytes - head, st)
                           : qn_fused_run(q, dtype, Wp, X, Y, row_idx, B, N, Nb, sse, pred, Gp,
                                          static_cast<char*>(ws) + head, ws_bytes - head, st);
    if (rc) return rc;
    if (gradW) {
        gx = (int)((d->p + 255) / 256);
        hipLaunchKernelGGL(k_unpad_grad<T>, dim3(gx > 64 ? 64 : gx, B), dim3(256), 0, st, m, (const T*)Gp, gradW);
    }
    {   // chains with unbounded values: once more from the original weights (returns at once for all others)
        FixNet f;
        f.L = L; f.has_bias = d->has_bias; f.act = d->act; f.d = d->dims[0]; f.o = d->dims[L]; f.p = d->p;
        int off = 0, hmax = 0;
        for (int l = 0; l <= L; ++l) f.dims[l] = d->dims[l];
        for (int l = 0; l < L; ++l) {
            f.offW[l] = d->offW[l]; f.offB[l] = d->offB[l];
            f.acto[l] = off; off += d->dims[l + 1];
            hmax = d->dims[l + 1] > hmax ? d->dims[l + 1] : hmax;
        }
        f.dzo = off; f.hmax = hmax; f.per_chain = (int)padded_fixup_doubles(d);
        hipLaunchKernelGGL(k_padded_fixup<T>, dim3(B), dim3(256), 0, st, f, W, static_cast<const T*>(X), static_cast<const T*>(Y),
                           row_idx, Nb, (const int*)flags, nslots, sse, static_cast<T*>(pred), gradW, fix_scratch);
    }
    QN_HIP_CHECK(hipGetLastError());
    return QN_OK;
}
int run_padded(const qn_desc* d, bool generic, int dtype, const void* W, const void* X, const void* Y,
               const int32_t* row_idx, int B, int N, int Nb, double* sse, void* pred, void* gradW, void* ws,
               size_t ws_bytes, hipStream_t st) {
    if (dtype == QN_F32)
        return run_padded_t<float>(d, generic, dtype, (const float*)W, X, Y, row_idx, B, N, Nb, sse, pred, (float*)gradW,
                                   ws, ws_bytes, st);
    return run_padded_t<double>(d, generic, dtype, (const double*)W, X, Y, row_idx, B, N, Nb, sse, pred, (double*)gradW,
                                ws, ws_bytes, st);
}
}  // namespace

static bool use_rnet_fused(const qn_desc* d, int want_grad, int dtype) {
    return d->kind == QN_KIND_RNET && d->path != QN_PATH_GENERIC && qn_rnet_fused_supported(d, want_grad, dtype);
}

extern "C" int qn_mlp_path(const qn_desc* d, int B, int Nb, int want_grad, int dtype) {
    if (!d) return QN_EINVAL;
    if (d->kind == QN_KIND_RNET) return use_rnet_fused(d, want_grad, dtype) ? QN_PATH_FUSED : QN_PATH_GENERIC;
    return use_fused(d, B, Nb, want_grad, dtype) ? QN_PATH_FUSED : QN_PATH_GENERIC;
}

extern "C" int qn_mlp_arith(const qn_desc* d, int B, int Nb, int want_grad, int dtype) {
    if (!d) return QN_EINVAL;
    if (d->kind != QN_KIND_MLP || dtype != QN_F64) return QN_ARITH_PLAIN;
    if (use_fused(d, B, Nb, want_grad, dtype)) return qn_fused_uses_i8(fused_desc(d), want_grad) ? QN_ARITH_I8_FUSED : QN_ARITH_PLAIN;
    if (d->path != QN_PATH_AUTO) return QN_ARITH_PLAIN;
    const qn_desc* g = use_padded_generic(d) ? d->padded : d;                 // the network the layer-wise family runs
    if (qn_i8_wide_applies(g)) return QN_ARITH_I8_WIDE;
    return qn_i8_layers_apply(g) ? QN_ARITH_I8_LAYERS : QN_ARITH_PLAIN;
}

extern "C" size_t qn_workspace_bytes(const qn_desc* d, int B, int Nb, int want_grad, int dtype) {
    if (!d || B <= 0 || Nb <= 0) return 0;
    if (d->kind == QN_KIND_RNET) {      // sized for either family so that qn_mlp_desc_set_path never invalidates a caller's buffer
        const size_t g = qn_rnet_workspace(d, B, Nb, want_grad, dtype);
        const size_t f = qn_rnet_fused_supported(d, want_grad, dtype) ? qn_rnet_fused_workspace(d, B, Nb, want_grad) : 0;
        return g > f ? g : f;
    }
    // sized for either family so that qn_mlp_desc_set_path never invalidates a caller's buffer
    size_t g = qn_generic_workspace(d, B, Nb, want_grad, dtype);
    size_t f = fused_ok(d, B, Nb, want_grad, dtype) ? fused_ws(d, B, Nb, want_grad, dtype) : 0;
    if (d->path == QN_PATH_AUTO && f && !prefer_wide(d, want_grad, dtype)) return f;
    if (use_padded_generic(d)) {
        const size_t pg = padded_generic_ws(d, B, Nb, want_grad, dtype);
        g = pg > g ? pg : g;
    }
    return g > f ? g : f;
}

static int check_common(const char* fn, const qn_desc* d, int dtype, const void* W, const void* X, const void* Y,
                        const int32_t* row_idx, int B, int N, int Nb, double* sse, void* ws) {
    if (!d || !W || !X || !Y || !sse || !ws) {
        qn_set_error("%s: null argument", fn);
        return QN_EINVAL;
    }
    if (dtype != QN_F64 && dtype != QN_F32) {
        qn_set_error("%s: dtype %d", fn, dtype);
        return QN_EINVAL;
    }
    if (B <= 0 || N <= 0 || Nb <= 0 || B > 65535) {
        qn_set_error("%s: B=%d N=%d Nb=%d out of range", fn, B, N, Nb);
        return QN_EINVAL;
    }
    if (!row_idx && Nb != N) {
        qn_set_error("%s: row_idx is NULL but Nb (%d) != N (%d)", fn, Nb, N);
        return QN_EINVAL;
    }
    return QN_OK;
}

static int run(const char* fn, const qn_desc* d, int dtype, const void* W, const void* X, const void* Y,
               const int32_t* row_idx, int B, int N, int Nb, double* sse, void* pred, void* gradW, void* ws,
               size_t ws_bytes, void* stream) {
    int rc = check_common(fn, d, dtype, W, X, Y, row_idx, B, N, Nb, sse, ws);
    if (rc) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int want_grad = gradW != nullptr;
    if (d->kind == QN_KIND_RNET) {
        if ((d->path == QN_PATH_FUSED || d->path == QN_PATH_FUSED_DP) && !qn_rnet_fused_supported(d, want_grad, dtype)) {
            qn_set_error("%s: fused path forced but not supported for this residual network", fn);
            return QN_EUNSUPPORTED;
        }
        if (use_rnet_fused(d, want_grad, dtype))
            return qn_rnet_fused_run(d, W, X, Y, row_idx, B, N, Nb, sse, pred, gradW, ws, ws_bytes, st);
        return qn_rnet_run(d, dtype, W, X, Y, row_idx, B, N, Nb, sse, pred, gradW, ws, ws_bytes, st);
    }
    if ((d->path == QN_PATH_FUSED || d->path == QN_PATH_FUSED_DP) && !fused_ok(d, B, Nb, want_grad, dtype)) {
        qn_set_error("%s: fused path forced but not supported for this shape", fn);
        return QN_EUNSUPPORTED;
    }
    if (use_fused(d, B, Nb, want_grad, dtype)) {
        if (d->padded)
            return run_padded(d, false, dtype, W, X, Y, row_idx, B, N, Nb, sse, pred, gradW, ws, ws_bytes, st);
        return qn_fused_run(d, dtype, W, X, Y, row_idx, B, N, Nb, sse, pred, gradW, ws, ws_bytes, st);
    }
    if (use_padded_generic(d))
        return run_padded(d, true, dtype, W, X, Y, row_idx, B, N, Nb, sse, pred, gradW, ws, ws_bytes, st);
    return qn_generic_run(d, dtype, W, X, Y, row_idx, B, N, Nb, sse, pred, gradW, ws, ws_bytes, st);
}

extern "C" int qn_mlp_sse_fwd(const qn_desc* d, int dtype, const void* W, const void* X, const void* Y,
                              const int32_t* row_idx, int B, int N, int Nb, double* sse_out, void* pred_out,
                              void* workspace, size_t workspace_bytes, void* stream) {
    return run("qn_mlp_sse_fwd", d, dtype, W, X, Y, row_idx, B, N, Nb, sse_out, pred_out, nullptr, workspace,
               workspace_bytes, stream);
}

extern "C" int qn_mlp_sse_parts(const qn_desc* d, int B, int Nb, int dtype) {
    if (!d || B <= 0 || Nb <= 0) return QN_EINVAL;
    if (d->kind == QN_KIND_MLP && !d->padded && use_fused(d, B, Nb, 0, dtype)) return qn_fused_parts(d, B, Nb);
    return 1;
}

extern "C" int qn_mlp_sse_fwd_parts(const qn_desc* d, int dtype, const void* W, const void* X, const void* Y,
                                    const int32_t* row_idx, int B, int N, int Nb, double* sse_parts_out,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    const int parts = qn_mlp_sse_parts(d, B, Nb, dtype);
    if (parts <= 1)
        return run("qn_mlp_sse_fwd_parts", d, dtype, W, X, Y, row_idx, B, N, Nb, sse_parts_out, nullptr, nullptr, workspace,
                   workspace_bytes, stream);
    if (int rc = check_common("qn_mlp_sse_fwd_parts", d, dtype, W, X, Y, row_idx, B, N, Nb, sse_parts_out, workspace)) return rc;
    return qn_fused_run(d, dtype, W, X, Y, row_idx, B, N, Nb, sse_parts_out, nullptr, nullptr, workspace, workspace_bytes,
                        static_cast<hipStream_t>(stream), true);
}

extern "C" int qn_mlp_sse_fwdbwd(const qn_desc* d, int dtype, const void* W, const void* X, const void* Y,
                                 const int32_t* row_idx, int B, int N, int Nb, double* sse_out, void* pred_out,
                                 void* gradW_out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!gradW_out) {
        qn_set_error("qn_mlp_sse_fwdbwd: gradW_out is NULL");
        return QN_EINVAL;
    }
    return run("qn_mlp_sse_fwdbwd", d, dtype, W, X, Y, row_idx, B, N, Nb, sse_out, pred_out, gradW_out, workspace,
               workspace_bytes, stream);
}

// ------------------------------------------------------------------------------------------
// VI: sampling + KL terms, chain rule to (mu, rho); batched Adam.  HBM-bound elementwise /
// row-reduction kernels: coalesced along the parameter index, fixed-order reductions.
namespace {

constexpr int BLK = 256;
constexpr double kLogSqrt2Pi = 0.91893853320467274178;   // log(sqrt(2*pi))

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// log N(w; 0, s) as torch.distributions.Normal.log_prob orders it:
//   -(w^2) / (2 s^2) - log(s) - log(sqrt(2 pi))
__device__ __forceinline__ double normal_logpdf(double w, double var2, double log_s) {
    return -(w * w) / var2 - log_s - kLogSqrt2Pi;
}

constexpr int KLB = 1024;   // one block per MC sample: fixed-order reduction, no scratch

template <typename T>
__global__ __launch_bounds__(KLB) void k_vi_sample_kl(const double* __restrict__ mu, const double* __restrict__ rho,
                                                      const double* __restrict__ eps, int64_t p, double pi,
                                                      double s1, double s2, T* __restrict__ Wout,
                                                      double* __restrict__ logq, double* __restrict__ logp) {
    __shared__ double red[2][KLB / 64];
    const int s = blockIdx.x;
    const double v1 = 2.0 * s1 * s1, v2 = 2.0 * s2 * s2, l1 = log(s1), l2 = log(s2);
    double aq = 0.0, ap = 0.0;
    for (int64_t i = threadIdx.x; i < p; i += KLB) {
        const double m = mu[i], r = rho[i], e = eps[(int64_t)s * p + i];
        const double sig = exp(r);
        const double w = m + sig * e;
        Wout[(int64_t)s * p + i] = (T)w;
        const double dq = w - m;
        aq += -kLogSqrt2Pi - r - (dq * dq) / (2.0 * (sig * sig));   // log(exp(r)) == r
        const double p1 = exp(normal_logpdf(w, v1, l1)), p2 = exp(normal_logpdf(w, v2, l2));
        ap += log(pi * p1 + (1.0 - pi) * p2);
    }
    aq = wave_sum(aq);
    ap = wave_sum(ap);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = aq; red[1][threadIdx.x >> 6] = ap; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double q = 0.0, pp = 0.0;
        for (int w = 0; w < KLB / 64; ++w) { q += red[0][w]; pp += red[1][w]; }
        logq[s] = q;
        logp[s] = pp;
    }
}

template <typename T>
__global__ __launch_bounds__(BLK) void k_vi_grad(const double* __restrict__ mu, const double* __restrict__ rho,
                                                 const double* __restrict__ eps, const T* __restrict__ gW, int S,
                                                 int64_t p, double pi, double s1, double s2, double gw_scale,
                                                 double kl_scale, double* __restrict__ dmu, double* __restrict__ drho) {
    const int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x;
    if (i >= p) return;
    const double m = mu[i], sig = exp(rho[i]);
    const double v1 = 2.0 * s1 * s1, v2 = 2.0 * s2 * s2, l1 = log(s1), l2 = log(s2);
    double gm = 0.0, gr = 0.0, pm = 0.0, pr = 0.0;
    for (int s = 0; s < S; ++s) {
        const double e = eps[(int64_t)s * p + i];
        const double w = m + sig * e;
        const double g = (double)gW[(int64_t)s * p + i] * gw_scale;
        gm += g;
        gr += g * sig * e;
        const double p1 = pi * exp(normal_logpdf(w, v1, l1)), p2 = (1.0 - pi) * exp(normal_logpdf(w, v2, l2));
        const double gp = (p1 * (-w / (s1 * s1)) + p2 * (-w / (s2 * s2))) / (p1 + p2);
        pm += gp;
        pr += gp * sig * e;
    }
    dmu[i] = gm - kl_scale * (pm / S);
    drho[i] = gr - kl_scale * (1.0 + pr / S);
}

template <typename T>
__global__ __launch_bounds__(BLK) void k_adam(double* __restrict__ W, const T* __restrict__ G, double* __restrict__ m,
                                              double* __restrict__ v, const double* __restrict__ lr, int64_t p,
                                              double gscale, double wd, double b1, double b2, double eps, double bc1,
                                              double bc2_sqrt) {
    const int b = blockIdx.y;
    const double lrb = lr[b];
    if (lrb == 0.0) return;
    const double step_size = lrb / bc1;
    for (int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x; i < p; i += (int64_t)gridDim.x * BLK) {
        const int64_t k = (int64_t)b * p + i;
        double w = W[k];
        double g = (double)G[k] * gscale;
        if (wd != 0.0) g = g + wd * w;
        double mm = m[k];
        mm = mm + (1.0 - b1) * (g - mm);                 // lerp_(grad, 1-beta1), weight < 0.5 branch
        double vv = v[k] * b2;
        vv = vv + ((1.0 - b2) * g) * g;                  // addcmul_(grad, grad, value=1-beta2)
        const double denom = sqrt(vv) / bc2_sqrt + eps;
        w = w + (-step_size) * (mm / denom);             // addcdiv_(exp_avg, denom, value=-step_size)
        W[k] = w;
        m[k] = mm;
        v[k] = vv;
    }
}

}  // namespace

extern "C" int qn_vi_sample_kl(const double* mu, const double* rho, const double* eps, int S, int64_t p,
                               double pi, double sigma1, double sigma2, int dtype, void* W_out, double* logq_out,
                               double* logp_out, void* stream) {
    if (!mu || !rho || !eps || !W_out || !logq_out || !logp_out || S <= 0 || p <= 0 || S > 65535) {
        qn_set_error("qn_vi_sample_kl: bad argument");
        return QN_EINVAL;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    (void)hipGetLastError();
    if (dtype == QN_F64)
        hipLaunchKernelGGL((k_vi_sample_kl<double>), dim3(S), dim3(KLB), 0, st, mu, rho, eps, p, pi, sigma1, sigma2,
                           (double*)W_out, logq_out, logp_out);
    else
        hipLaunchKernelGGL((k_vi_sample_kl<float>), dim3(S), dim3(KLB), 0, st, mu, rho, eps, p, pi, sigma1, sigma2,
                           (float*)W_out, logq_out, logp_out);
    QN_HIP_CHECK(hipGetLastError());
    return QN_OK;
}

extern "C" int qn_vi_grad(const double* mu, const double* rho, const double* eps, const void* gW, int S, int64_t p,
                          double pi, double sigma1, double sigma2, double gw_scale, double kl_scale, int dtype,
                          double* dmu_out, double* drho_out, void* stream) {
    if (!mu || !rho || !eps || !gW || !dmu_out || !drho_out || S <= 0 || p <= 0) {
        qn_set_error("qn_vi_grad: bad argument");
        return QN_EINVAL;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid((unsigned)((p + BLK - 1) / BLK));
    (void)hipGetLastError();
    if (dtype == QN_F64)
        hipLaunchKernelGGL((k_vi_grad<double>), grid, dim3(BLK), 0, st, mu, rho, eps, (const double*)gW, S, p, pi,
                           sigma1, sigma2, gw_scale, kl_scale, dmu_out, drho_out);
    else
        hipLaunchKernelGGL((k_vi_grad<float>), grid, dim3(BLK), 0, st, mu, rho, eps, (const float*)gW, S, p, pi,
                           sigma1, sigma2, gw_scale, kl_scale, dmu_out, drho_out);
    QN_HIP_CHECK(hipGetLastError());
    return QN_OK;
}

extern "C" int qn_adam_batched(double* W, const void* G, double* m, double* v, const double* lr, int B, int64_t p,
                               int dtype, double gscale, double wd, double beta1, double beta2, double eps, int step,
                               void* stream) {
    if (!W || !G || !m || !v || !lr || B <= 0 || B > 65535 || p <= 0 || step < 1) {
        qn_set_error("qn_adam_batched: bad argument");
        return QN_EINVAL;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const double bc1 = 1.0 - std::pow(beta1, (double)step);
    const double bc2_sqrt = std::sqrt(1.0 - std::pow(beta2, (double)step));
    int nblk = (int)((p + BLK - 1) / BLK);
    if (nblk > 1024) nblk = 1024;
    dim3 grid(nblk, B);
    (void)hipGetLastError();
    if (dtype == QN_F64)
        hipLaunchKernelGGL((k_adam<double>), grid, dim3(BLK), 0, st, W, (const double*)G, m, v, lr, p, gscale, wd,
                           beta1, beta2, eps, bc1, bc2_sqrt);
    else
        hipLaunchKernelGGL((k_adam<float>), grid, dim3(BLK), 0, st, W, (const float*)G, m, v, lr, p, gscale, wd,
                           beta1, beta2, eps, bc1, bc2_sqrt);
    QN_HIP_CHECK(hipGetLastError());
    return QN_OK;
}

// ------------------------------------------------------------------------------------------
// Diagnostic: the device tanh on an array (accuracy tests of qn_math.h).
namespace {
template <bool NANSAFE>
__global__ void k_tanh_f64(const double* __restrict__ x, double* __restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = qn_tanh_f64_impl<NANSAFE>(x[i]);
}
template <bool NANSAFE>
__global__ void k_tanh_f64_tab(const double* __restrict__ x, double* __restrict__ y, int64_t n) {
    __shared__ double tab[QN_TANH_LDS_DOUBLES];
    qn_tanh_table_stage(tab, threadIdx.x, blockDim.x);
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = qn_tanh_f64_tab<NANSAFE>(x[i], tab);
}
__global__ void k_tanh_f64_tab64(const double* __restrict__ x, double* __restrict__ y, int64_t n) {
    __shared__ double tab[QN_TANH64_LDS_DOUBLES];
    qn_tanh_table64_stage(tab, threadIdx.x, blockDim.x);
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = qn_tanh_f64_tab64(x[i], tab);
}
int debug_tanh(const char* fn, int variant, const double* x, double* y, int64_t n, void* stream) {
    if (!x || !y || n <= 0) {
        qn_set_error("%s: bad argument", fn);
        return QN_EINVAL;
    }
    (void)hipGetLastError();
    const dim3 grid((unsigned)((n + 255) / 256));
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (variant == 0) hipLaunchKernelGGL(k_tanh_f64<true>, grid, dim3(256), 0, st, x, y, n);
    else if (variant == 1) hipLaunchKernelGGL(k_tanh_f64<false>, grid, dim3(256), 0, st, x, y, n);
    else if (variant == 2) hipLaunchKernelGGL(k_tanh_f64_tab<true>, grid, dim3(256), 0, st, x, y, n);
    else if (variant == 3) hipLaunchKernelGGL(k_tanh_f64_tab<false>, grid, dim3(256), 0, st, x, y, n);
    else hipLaunchKernelGGL(k_tanh_f64_tab64, grid, dim3(256), 0, st, x, y, n);
    QN_HIP_CHECK(hipGetLastError());
    return QN_OK;
}
}  // namespace

extern "C" int qn_debug_tanh(const double* x, double* y, int64_t n, void* stream) {
    return debug_tanh("qn_debug_tanh", 0, x, y, n, stream);
}
extern "C" int qn_debug_tanh_finite(const double* x, double* y, int64_t n, void* stream) {
    return debug_tanh("qn_debug_tanh_finite", 1, x, y, n, stream);
}
extern "C" int qn_debug_tanh_table(const double* x, double* y, int64_t n, int nansafe, void* stream) {
    return debug_tanh("qn_debug_tanh_table", nansafe == 2 ? 4 : (nansafe ? 2 : 3), x, y, n, stream);
}

// ---------------------------------------------------------------------------------------------- predictive moments
// mean / unbiased variance over the M members of a predictive ensemble Y [M, K] (K = N * o columns), per column, in
// float64 and in the member order numpy uses for axis 0 (np.mean, np.var(ddof=1) of quinn/solvers/quinn.py:93-99):
// one thread per column, two passes over its M values, coalesced across columns.  HBM-bound: 2 x M x K reads.
namespace {
template <typename T>
__global__ __launch_bounds__(256) void k_pred_moments(const T* __restrict__ Y, int64_t M, int64_t K, double* __restrict__ mean,
                                                      double* __restrict__ var) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= K) return;
    double s = 0.0;
    for (int64_t m = 0; m < M; ++m) s += (double)Y[m * K + j];
    const double mu = s / (double)M;
    mean[j] = mu;
    if (var) {
        double q = 0.0;
        for (int64_t m = 0; m < M; ++m) {
            const double d = (double)Y[m * K + j] - mu;
            q += d * d;
        }
        var[j] = q / (double)(M - 1);
    }
}
}  // namespace

extern "C" int qn_pred_moments(const void* Y, int dtype, int64_t M, int64_t K, double* mean_out, double* var_out, void* stream) {
    if (!Y || !mean_out || M < 1 || K < 1 || (var_out && M < 2) || (dtype != QN_F64 && dtype != QN_F32)) {
        qn_set_error("qn_pred_moments: bad argument");
        return QN_EINVAL;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    (void)hipGetLastError();
    const dim3 grid((unsigned)((K + 255) / 256));
    if (dtype == QN_F32) hipLaunchKernelGGL(k_pred_moments<float>, grid, dim3(256), 0, st, (const float*)Y, M, K, mean_out, var_out);
    else hipLaunchKernelGGL(k_pred_moments<double>, grid, dim3(256), 0, st, (const double*)Y, M, K, mean_out, var_out);
    QN_HIP_CHECK(hipGetLastError());
    return QN_OK;
}
