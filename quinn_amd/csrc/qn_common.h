// Internal definitions shared by the HIP translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include <vector>
#include "../../include/quinn_amd.h"

#define QN_MAX_LAYERS 16

struct qn_desc {
    int nlayers;                    // number of Linear layers (= ndims - 1)
    int dims[QN_MAX_LAYERS + 1];    // d, h_1, ..., o
    int act;
    int has_bias;
    int64_t p;                      // flat parameter count
    int64_t offW[QN_MAX_LAYERS];    // offset of W_l in the flat vector
    int64_t offB[QN_MAX_LAYERS];    // offset of b_l (valid if has_bias)
    int hmax;                       // widest hidden/output layer
    // ---- residual network (quinn/nns/rnet.py:16-170); kind == QN_KIND_RNET.  dims[] = {d, r, o}.
    int kind;
    int rn_r, rn_steps, rn_npar, rn_pre, rn_post, rn_mlp;
    unsigned char rn_uses[QN_MAX_LAYERS * QN_MAX_LAYERS];   // [steps][npar]: tensor k enters step i at all (qn_rnet_desc_set_uses); a
                                    // tensor that does not is SKIPPED, not multiplied by its zero coefficient (0 . Inf = NaN)
    int64_t rn_offWpre, rn_offBpre, rn_offWpost, rn_offBpost, rn_offWW, rn_offBB;
    double rn_coef[QN_MAX_LAYERS * QN_MAX_LAYERS];   // [steps][npar]: W_i = sum_k coef[i][k] * ww_k
    // ---- MLP whose hidden widths are <= 64 but not all equal to 16 / 32 / 64: the same network with every hidden
    // layer zero-padded to one of those widths, which the fused kernels take (qn_api.hip: pad -> fused -> unpad)
    qn_desc* padded;
    // ---- kernel family forced on this descriptor (qn_mlp_desc_set_path; QN_PATH_AUTO = dispatch by shape)
    int path;
    // ---- the fused kernels split a chain's rows as a launch of max(B, plan_batch) chains would (qn_mlp_desc_set_plan_batch):
    // a batch run as several smaller launches then sums every chain's rows in the same order as the whole batch in one
    int plan_batch;
};
enum { QN_KIND_MLP = 0, QN_KIND_RNET = 1 };

void qn_set_error(const char* fmt, ...);

#define QN_HIP_CHECK(expr)                                                          \
    do {                                                                            \
        hipError_t e_ = (expr);                                                     \
        if (e_ != hipSuccess) {                                                     \
            qn_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),     \
                         __FILE__, __LINE__);                                       \
            return QN_EHIP;                                                         \
        }                                                                           \
    } while (0)

static inline size_t qn_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// ---- generic (any-shape) path: qn_generic.hip
size_t qn_generic_workspace(const qn_desc* d, int B, int Nb, int want_grad, int dtype);
int qn_generic_run(const qn_desc* d, int dtype, const void* W, const void* X, const void* Y,
                   const int32_t* row_idx, int B, int N, int Nb, double* sse, void* pred,
                   void* gradW, void* ws, size_t ws_bytes, hipStream_t st);

// ---- residual network, layer-wise: qn_generic.hip
size_t qn_rnet_workspace(const qn_desc* d, int B, int Nb, int want_grad, int dtype);
int qn_rnet_run(const qn_desc* d, int dtype, const void* W, const void* X, const void* Y,
                const int32_t* row_idx, int B, int N, int Nb, double* sse, void* pred,
                void* gradW, void* ws, size_t ws_bytes, hipStream_t st);

// ---- small residual networks, one launch: qn_rnet.hip
bool qn_rnet_fused_supported(const qn_desc* d, int want_grad, int dtype);
size_t qn_rnet_fused_workspace(const qn_desc* d, int B, int Nb, int want_grad);
int qn_rnet_fused_run(const qn_desc* d, const void* W, const void* X, const void* Y, const int32_t* row_idx, int B, int N,
                      int Nb, double* sse, void* pred, void* gradW, void* ws, size_t ws_bytes, hipStream_t st);

// ---- fused MFMA path with LDS-resident weights: qn_fused.hip
bool qn_fused_supported(const qn_desc* d, int B, int Nb, int want_grad, int dtype);
size_t qn_fused_workspace(const qn_desc* d, int B, int Nb, int want_grad, int dtype);
// parts_out (forward only): `sse` is [B][qn_fused_parts()] and receives the per-row-split partial sums (their plain
// left-to-right sum is the SSE); the final summation kernel is skipped
int qn_fused_run(const qn_desc* d, int dtype, const void* W, const void* X, const void* Y,
                 const int32_t* row_idx, int B, int N, int Nb, double* sse, void* pred,
                 void* gradW, void* ws, size_t ws_bytes, hipStream_t st, bool parts_out = false);
int qn_fused_parts(const qn_desc* d, int B, int Nb);
bool qn_fused_uses_i8(const qn_desc* d, int want_grad);     // the sliced int8-product kernels would run (not the float64-MFMA ones)
