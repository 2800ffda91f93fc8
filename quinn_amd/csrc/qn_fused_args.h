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

// ---- the SSE of a chain from the partial sums of its row splits, inside the forward kernel ----------------------------
// The LAST workgroup of a chain to finish adds the chain's nsplit partial sums (left to right: bit for bit what the
// separate k_sum_partials launch wrote): one dependent launch fewer per log-posterior step (BASELINE configs[1]: ~2 us of
// a 82 us step at a settled clock; the 14 us the round-2 bench line showed between step and kernel were mostly the GPU clock
// still ramping up during the timed region, profiles/r03_*).
//   arrive[b]: {QN_ARRIVE_MAGIC : 48 | arrivals of the running launch : 16}.  Anything else (a fresh workspace, bytes another
//   call left there) counts as "no arrival yet", so the caller has nothing to initialise; the last arriver leaves
//   {MAGIC, 0}.  (Workspace garbage that happens to equal MAGIC in its top 48 bits -- 2^-48 per chain and first use --
//   would be taken for a count.)
//   Memory order: partial[] and arrive[] are only touched with agent-scope atomics, which are performed at the agent's
//   coherence point, never in a non-coherent cache; a workgroup's partial-sum store is acknowledged (s_waitcnt vmcnt(0))
//   before its arrival is issued, and the last arriver's loads depend on the value its arrival returned.  (An agent-scope
//   RELEASE / ACQUIRE pair on the arrival adds an L2 write-back and an invalidate per workgroup: measured 3.5 us per launch
//   SLOWER than the separate launch, A/B in one call; this protocol: 1.5-2 us faster than the separate launch.)
#ifdef __HIPCC__
constexpr unsigned long long QN_ARRIVE_MAGIC = 0xFFF751A7C0DEull;           // (top 48 bits of a quiet-NaN pattern with a payload)
// One arrival at the tagged counter arrive[b] (call from ONE lane): returns how many arrivals of the running launch the slot
// has seen, this one included.  The last arriver resets the slot to {MAGIC, 0} when it is done.
__device__ __forceinline__ unsigned qn_arrive_tagged(unsigned long long* __restrict__ arrive, int b) {
    unsigned long long old = __hip_atomic_fetch_add(&arrive[b], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((old >> 16) == QN_ARRIVE_MAGIC) return (unsigned)(old & 0xffff) + 1;
    // first use of this slot (or bytes of another call): claim it; arrivals that raced with the claim retry against it
    unsigned long long cur = __hip_atomic_load(&arrive[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        const unsigned long long next = (cur >> 16) == QN_ARRIVE_MAGIC ? cur + 1 : ((QN_ARRIVE_MAGIC << 16) | 1ull);
        if (__hip_atomic_compare_exchange_strong(&arrive[b], &cur, next, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            return (unsigned)(next & 0xffff);
    }
}
// Called by ALL lanes of the workgroup's first wave (`value`: the workgroup's partial sum, the same in every lane).
__device__ __forceinline__ void qn_sse_finish(double* __restrict__ partial, unsigned long long* __restrict__ arrive,
                                              double* __restrict__ sse, int b, int split, int nsplit, double value) {
    const int lane = threadIdx.x & 63;
    if (!arrive) {                                                        // the caller sums (qn_mlp_sse_fwd_parts)
        if (lane == 0) partial[(int64_t)b * nsplit + split] = value;
        return;
    }
    unsigned count = 0;
    if (lane == 0) {
        __hip_atomic_store(&partial[(int64_t)b * nsplit + split], value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // vmcnt(0): the store is acknowledged before the arrival is issued.  The two relaxed atomics are on different addresses,
        // so the language gives the compiler no order between them: the signal fences pin it (compiler-only, no instruction).
        // gfx950-specific: the hardware order rests on the sc1 write-through store being complete when vmcnt reaches 0.
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        count = qn_arrive_tagged(arrive, b);
    }
    count = (unsigned)__builtin_amdgcn_readfirstlane((int)count);
    if (count != (unsigned)nsplit) return;
    // the last workgroup of the chain: one load per lane (ONE round trip for up to 64 parts), summed left to right
    double s = 0.0;
    for (int base = 0; base < nsplit; base += 64) {
        const int i = base + lane;
        const double v = i < nsplit ? __hip_atomic_load(&partial[(int64_t)b * nsplit + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        const int n = nsplit - base < 64 ? nsplit - base : 64;
        for (int k = 0; k < n; ++k) s += __shfl(v, k, 64);
    }
    if (lane == 0) {
        sse[b] = s;
        __hip_atomic_store(&arrive[b], QN_ARRIVE_MAGIC << 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
#endif

using qn_fwd_fn = void (*)(FusedArgs, const double*, const double*, const double*, const int32_t*, double*, double*,
                           unsigned long long*, double*);

// float64-MFMA fused kernels for networks with 5..16 inputs (qn_fused_d8.hip, the second object of qn_fused.hip): the gradient
// kernel k_fused_bwd_f64<H, NH, dp, UNB> (dp = 8, 16) and the relu / identity forward
// k_fused_fwd_f64<H, G, ACT, dp> (dp = 8, 16); null = no such instance
using qn_bwd_f64_fn = void (*)(FusedArgs, const double*, const double*, const double*, const int32_t*, double*, double*,
                               double*, const int*, double*, double*, unsigned long long*);
qn_bwd_f64_fn qn_fused_bwd_d8_kernel(int H, int nhid, int act, int dp);
qn_fwd_fn qn_fused_fwd_d8_kernel(int H, int act, int dp, int wide_out);
// the gradient kernel for networks with 5..16 outputs (qn_fused_o16.hip, the third object): dp = 4 or 16
qn_bwd_f64_fn qn_fused_bwd_o16_kernel(int H, int nhid, int act, int dp);

// sliced int8-product forward for 64-wide tanh networks (qn_fused_i8.hip): same grid, block and partial-sum
// conventions as k_fused_fwd_f64<64, 2, tanh, DP, 256>
int qn_fused_i8_rows_per_iteration();       // data rows one workgroup covers per loop iteration
bool qn_fused_i8_applies(int H, int nhid, int act, int d, int o);
size_t qn_fused_i8_lds_bytes(int d, int nhid);
qn_fwd_fn qn_fused_i8_kernel(int d, int o, int act);

// sliced int8-product forward + backward for 64-wide tanh networks (qn_fused_bwd_i8.hip): grid, partial-sum and gradient-slab
// conventions of k_fused_bwd_f64<64, NH, 4>; `flags` [B][nsplit]: (chain, split)s that left the fast path (the caller
// reruns those chains with k_fused_bwd_f64)
using qn_bwd_i8_fn = void (*)(FusedArgs, const double*, const double*, const double*, const int32_t*, double*, double*, double*, int*);
bool qn_fused_bwd_i8_applies(int H, int nhid, int act, int d, int o);
size_t qn_fused_bwd_i8_lds_bytes(int nhid, int d);
qn_bwd_i8_fn qn_fused_bwd_i8_kernel(int nhid, int d, int act);

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
// (Ns >= Nb: the row stride of the stashes [B][h][Ns] -- activations here, dZ in the backward: qn_generic.hip pads it to whole
// 128-byte lines)
int qn_i8_wide_forward(const qn_desc* d, const double* W, const double* X, const double* Y, const int32_t* row_idx, int B,
                       int Nb, double* act0, int64_t act_stride, int Ns, double* dz_last, double* pred, double* sse, void* ws,
                       hipStream_t st);
// the backward pass through the hidden layers of the same networks: dZ_l (float64 [B][h][Nb] at dz0 + l * dz_stride,
// l = 0 .. L-2) from the stashed activations and dz_last; same workspace as the forward call of the evaluation
int qn_i8_wide_backward(const qn_desc* d, const double* W, const double* X, const int32_t* row_idx, int B, int Nb,
                        const double* act0, int64_t act_stride, int Ns, const double* dz_last, double* dz0, int64_t dz_stride, void* ws,
                        double* gradW, int* last_done, hipStream_t st);
// weight gradient of a hidden->hidden layer as sliced int8 products (qn_dw_i8.hip): output conventions of k_gemm64<DW>
// rowsc (may be null: tanh, activations in [-1, 1]): [B][Nb] scales 2^f_n > every |a_prev[.][n]| of data row n, as the forward
// of a relu / identity network leaves them (qn_i8_wide_rowscale); with it, row counts in whole 64-row chunks only
int qn_i8_dw(int h_in, int h_out, int has_bias, const double* dz, const double* a_prev, int B, int Nb, int Ns, double* dst,
             int64_t out_stride_b, int64_t out_stride_k, int ksplit, int kchunk, const double* rowsc, hipStream_t st);
// the last Nb % 64 rows of the same product in float64, added to the finished gradient block (after the split-K reduction)
int qn_i8_dw_tail(int h_in, int h_out, int has_bias, const double* dz, const double* a_prev, int B, int Nb, int Ns, double* G, int64_t p,
                  hipStream_t st);
// relu / identity networks, gradient calls: the forward's per-row activation scales [L-2 layers][B][Nb] inside its workspace
// (layer l = the scales of act0 + l * act_stride); null for tanh networks / forward-only workspaces
double* qn_i8_wide_rowscale(const qn_desc* d, int B, int Nb, int want_grad, void* ws);
