// Launch arguments shared by the fused forward kernels (qn_fused.hip, qn_fused_i8.hip); not part of the C ABI.
#pragma once
#include <cstddef>
#include <cstdint>

struct FusedArgs {
    int64_t p;
    int B, N, Nb, d, o, nhid, act, has_bias;
    int nsplit, rows_per_split, iters;
    int64_t dbg_off;     // diagnostic builds: offset (doubles, from the partials) of a 12-word scratch
};

// Grid of the fused kernels: 1-D and XCD-aware.  Consecutive workgroup ids are dealt round-robin to the 8 XCDs (each
// with its own L2), so workgroup id -> (chain, row split) is chosen such that all splits of a chain share id % 8: the
// chain's weights come from HBM once and from that XCD's L2 for the other splits (grid (nsplit, B) put the 8 splits of a
// cfg2 chain on 8 different XCDs: 8 x the weight traffic, taken as one burst at kernel start).
inline unsigned qn_fused_grid(int nsplit, int B) { return (unsigned)(((B + 7) / 8) * 8 * nsplit); }
#ifdef __HIPCC__
__device__ __forceinline__ bool qn_fused_wg(int nsplit, int B, int* b, int* split) {
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    *b = (seq / nsplit) * 8 + xcd;
    *split = seq % nsplit;
    return *b < B;
}
#endif

using qn_fwd_fn = void (*)(FusedArgs, const double*, const double*, const double*, const int32_t*, double*, double*);

// sliced int8-product forward for 64-wide tanh networks (qn_fused_i8.hip): same grid, block and partial-sum
// conventions as k_fused_fwd_f64<64, 2, tanh, DP, 256>
int qn_fused_i8_rows_per_iteration();       // data rows one workgroup covers per loop iteration
bool qn_fused_i8_applies(int H, int nhid, int act, int d, int o);
size_t qn_fused_i8_lds_bytes(int d, int nhid);
qn_fwd_fn qn_fused_i8_kernel(int d, int o);

// sliced int8-product forward + backward for 64-wide tanh networks (qn_fused_bwd_i8.hip): grid, partial-sum and gradient-slab
// conventions of k_fused_bwd_f64<64, NH, 4>; `flags` [B][nsplit]: (chain, split)s that left the fast path (the caller
// reruns those chains with k_fused_bwd_f64)
using qn_bwd_i8_fn = void (*)(FusedArgs, const double*, const double*, const double*, const int32_t*, double*, double*, double*, int*);
bool qn_fused_bwd_i8_applies(int H, int nhid, int act, int d, int o);
size_t qn_fused_bwd_i8_lds_bytes(int nhid);
qn_bwd_i8_fn qn_fused_bwd_i8_kernel(int nhid, int d);

// layer-wise int8-slice forward for wide tanh networks (qn_fused_i8.hip); used by qn_generic.hip
struct qn_desc;
#include <hip/hip_runtime.h>
bool qn_i8_layers_apply(const qn_desc* d);
size_t qn_i8_layers_workspace(const qn_desc* d, int B, int Nb);
int qn_i8_layers_forward(const qn_desc* d, const double* W, const double* X, const int32_t* row_idx, int B, int Nb,
                         double* const* act, void* ws, hipStream_t st);

// fused int8-slice forward for 128 / 256-wide tanh networks (qn_wide_i8.hip); used by qn_generic.hip: one launch for the
// whole forward pass (sse, optional pred / dz_last = 2 (pred - y) / float64 hidden activations act0 + l * act_stride)
bool qn_i8_wide_applies(const qn_desc* d);
size_t qn_i8_wide_workspace(const qn_desc* d, int B, int Nb, int want_grad);
int qn_i8_wide_forward(const qn_desc* d, const double* W, const double* X, const double* Y, const int32_t* row_idx, int B,
                       int Nb, double* act0, int64_t act_stride, double* dz_last, double* pred, double* sse, void* ws,
                       hipStream_t st);
// the backward pass through the hidden layers of the same networks: dZ_l (float64 [B][h][Nb] at dz0 + l * dz_stride,
// l = 0 .. L-2) from the stashed activations and dz_last; same workspace as the forward call of the evaluation
int qn_i8_wide_backward(const qn_desc* d, const double* W, const double* X, const int32_t* row_idx, int B, int Nb,
                        const double* act0, int64_t act_stride, const double* dz_last, double* dz0, int64_t dz_stride, void* ws,
                        hipStream_t st);
// weight gradient of a hidden->hidden layer as sliced int8 products (qn_dw_i8.hip): output conventions of k_gemm64<DW>
int qn_i8_dw(int h_in, int h_out, int has_bias, const double* dz, const double* a_prev, int B, int Nb, double* dst,
             int64_t out_stride_b, int64_t out_stride_k, int ksplit, int kchunk, hipStream_t st);
