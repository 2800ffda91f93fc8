// Single-launch kernels for SMALL residual networks (quinn/nns/rnet.py:130-165; rdim <= 8, the sizes the reference
// uses: 3 in examples/ex_ufit.py, 3-5 in tests/test_mlp.py).  The layer-wise path (qn_generic.hip: run_rnet) needs
// ~13 launches for a forward and ~35 for a gradient and streams every per-step activation through HBM; here one
// thread carries one data row through all steps in registers.
//   weights   : the chain's step weights W_i = sum_k coef[i][k] ww_k are expanded by every block into LDS
//               (a few hundred doubles) and read as wave-uniform broadcasts;
//   forward   : out (R doubles) in registers, tanh by the table-assisted routine (qn_math.h);
//   backward  : the per-step states out_i / tanh_i of the thread's row go to an LDS stash [state][thread]
//               (consecutive threads -> consecutive words: conflict-free), the per-step weight gradients are
//               accumulated thread-privately in LDS [step][entry][thread] over all rows of the thread (dynamic step
//               index without scratch), pre / post layer gradients in registers; one block reduction at the end,
//               per-block partials in a slab, k_rnet_grad_reduce sums the blocks in a fixed order and contracts the
//               step gradients back with the coefficients (d/d ww_k = sum_i coef[i][k] d/d W_i).
// Threads per block are chosen so that stash + accumulators fit LDS (256 / 128 / 64); networks that do not fit
// fall back to the layer-wise path.  float64 only.  Bound: the DP VALU (tanh) -- per row ~S*R tanh and 2 S R^2 flops.
#include "qn_common.h"
#include "qn_math.h"
#include <algorithm>

namespace {

constexpr int RMAX = 8, DOMAX = 4, DOWIDE = 16;     // inputs / outputs: 4 (forward + backward), 16 (forward kernel only)

struct RnFusedArgs {
    int64_t p;                      // flat parameters per chain
    int B, N, Nb, d, o, r, S, npar, pre, post, mlp, has_bias, act;
    int64_t offWpre, offBpre, offWpost, offBpost, offWW, offBB;
    int T;                          // threads per block (backward: also the stash stride)
    int nblk;                       // row blocks per chain (grid.x)
    double coef[QN_MAX_LAYERS * QN_MAX_LAYERS];
    unsigned char use[QN_MAX_LAYERS * QN_MAX_LAYERS];   // tensor k enters step i (qn_common.h: rn_uses): others are skipped
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// LDS image (doubles): Wpre [R][DOMAX] | bpre [R] | Weff [S][R*R + R] | Wpost [DOMAX][R] | bpost [DOMAX] | tanh table
template <int R, int DO = DOMAX> __host__ __device__ constexpr int img_doubles(int S) {
    return R * DO + R + S * (R * R + R) + DO * R + DO;
}

template <int R, int DO = DOMAX>
__device__ __forceinline__ void stage(const RnFusedArgs& a, const double* __restrict__ Wb, double* lds, int tid, int nt) {
    const int r = a.r;
    double* Wpre = lds;
    double* bpre = Wpre + R * DO;
    double* Weff = bpre + R;
    double* Wpost = Weff + a.S * (R * R + R);
    double* bpost = Wpost + DO * R;
    for (int e = tid; e < R * DO; e += nt) {
        const int j = e / DO, k = e % DO;
        Wpre[e] = (a.pre && j < r && k < a.d) ? Wb[a.offWpre + j * a.d + k] : 0.0;
    }
    for (int e = tid; e < R; e += nt) bpre[e] = (a.pre && e < r) ? Wb[a.offBpre + e] : 0.0;
    const int per = R * R + R;
    for (int e = tid; e < a.S * per; e += nt) {
        const int i = e / per, q = e % per;
        double s = 0.0;
        if (q < R * R) {
            const int j = q / R, k = q % R;
            if (j < r && k < r)
                for (int m = 0; m < a.npar; ++m)
                    if (a.use[i * a.npar + m]) s = fma(a.coef[i * a.npar + m], Wb[a.offWW + (int64_t)m * r * r + j * r + k], s);
        } else if (a.has_bias) {
            const int j = q - R * R;
            if (j < r)
                for (int m = 0; m < a.npar; ++m)
                    if (a.use[i * a.npar + m]) s = fma(a.coef[i * a.npar + m], Wb[a.offBB + (int64_t)m * r + j], s);
        }
        Weff[e] = s;
    }
    for (int e = tid; e < DO * R; e += nt) {
        const int q = e / R, k = e % R;
        Wpost[e] = (a.post && q < a.o && k < r) ? Wb[a.offWpost + q * r + k] : 0.0;
    }
    for (int e = tid; e < DO; e += nt) bpost[e] = (a.post && e < a.o) ? Wb[a.offBpost + e] : 0.0;
}

__device__ __forceinline__ double act_f(double z, int act, const double* tab) {
    return act == QN_ACT_TANH ? qn_tanh_f64_tab<true>(z, tab) : z;
}
__device__ __forceinline__ double act_d(double a, int act) { return act == QN_ACT_TANH ? 1.0 - a * a : 1.0; }

// one data row through the network; states written to the stash when GRAD
template <int R, bool GRAD, int DO = DOMAX>
__device__ __forceinline__ void forward_row(const RnFusedArgs& a, const double* lds, const double* tab, const double (&x)[DO],
                                            double (&out)[R], double* stash, int T) {
    const double* Wpre = lds;
    const double* bpre = Wpre + R * DO;
    const double* Weff = bpre + R;
    const double h = 1.0 / a.S;
    if (a.pre) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            double z = bpre[j];
#pragma unroll
            for (int k = 0; k < DO; ++k) z = fma(Wpre[j * DO + k], x[k], z);
            out[j] = j < a.r ? act_f(z, a.act, tab) : 0.0;
        }
    } else {
#pragma unroll
        for (int j = 0; j < R; ++j) out[j] = (j < DO && j < a.r) ? x[j < DO ? j : 0] : 0.0;
    }
    for (int i = 0; i < a.S; ++i) {
        const double* Wi = Weff + i * (R * R + R);
        double th[R];
#pragma unroll
        for (int j = 0; j < R; ++j) {
            double z = Wi[R * R + j];
#pragma unroll
            for (int k = 0; k < R; ++k) z = fma(Wi[j * R + k], out[k], z);
            th[j] = j < a.r ? act_f(z, a.act, tab) : 0.0;
        }
        if (GRAD) {
#pragma unroll
            for (int j = 0; j < R; ++j)
                if (j < a.r) {
                    stash[(size_t)((2 * i) * a.r + j) * T] = out[j];          // out_i
                    stash[(size_t)((2 * i + 1) * a.r + j) * T] = th[j];       // act(z_i)
                }
        }
#pragma unroll
        for (int j = 0; j < R; ++j) out[j] = a.mlp ? th[j] : fma(h, th[j], out[j]);
    }
}

// DO = 4, or 16 for networks with more than 4 inputs / outputs (forward only)
template <int R, int DO = DOMAX>
__global__ __launch_bounds__(256) void k_rnet_fwd(RnFusedArgs a, const double* __restrict__ W, const double* __restrict__ X,
                                                  const double* __restrict__ Y, const int32_t* __restrict__ row_idx,
                                                  double* __restrict__ pred, double* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* lds = reinterpret_cast<double*>(smem);
    double* tab = lds + ((img_doubles<R, DO>(a.S) + 1) & ~1);
    double* red = tab + QN_TANH_LDS_DOUBLES + 1;
    const int b = blockIdx.y, tid = threadIdx.x;
    stage<R, DO>(a, W + (int64_t)b * a.p, lds, tid, blockDim.x);
    qn_tanh_table_stage(tab, tid, blockDim.x);
    __syncthreads();
    const double* Wpost = lds + R * DO + R + a.S * (R * R + R);
    const double* bpost = Wpost + DO * R;
    double sse = 0.0;
    for (int n = blockIdx.x * blockDim.x + tid; n < a.Nb; n += gridDim.x * blockDim.x) {
        const int64_t row = row_idx ? (int64_t)row_idx[(int64_t)b * a.Nb + n] : (int64_t)n;
        double x[DO];
#pragma unroll
        for (int k = 0; k < DO; ++k) x[k] = k < a.d ? X[row * a.d + k] : 0.0;
        double out[R];
        forward_row<R, false, DO>(a, lds, tab, x, out, nullptr, 0);
#pragma unroll
        for (int q = 0; q < DO; ++q) {
            if (q >= a.o) break;
            double pr;
            if (a.post) {
                pr = bpost[q];
#pragma unroll
                for (int k = 0; k < R; ++k) pr = fma(Wpost[q * R + k], out[k], pr);
            } else {
                pr = out[q < R ? q : 0];
            }
            const double res = pr - Y[row * a.o + q];
            sse += res * res;
            if (pred) pred[((int64_t)b * a.Nb + n) * a.o + q] = pr;
        }
    }
    sse = wave_sum(sse);
    if ((tid & 63) == 0) red[tid >> 6] = sse;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int w = 0; w < (int)blockDim.x / 64; ++w) s += red[w];
        partial[(int64_t)b * a.nblk + blockIdx.x] = s;
    }
}

// effective-gradient layout of one block's partial (doubles): dWpre [R][DOMAX] | dbpre [R] | dWeff [S][R*R+R] |
// dWpost [DOMAX][R] | dbpost [DOMAX]   ( = the LDS image layout, img_doubles<R>(S) entries)
template <int R>
__global__ __launch_bounds__(256) void k_rnet_bwd(RnFusedArgs a, const double* __restrict__ W, const double* __restrict__ X,
                                                  const double* __restrict__ Y, const int32_t* __restrict__ row_idx,
                                                  double* __restrict__ pred, double* __restrict__ partial,
                                                  double* __restrict__ slab) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* lds = reinterpret_cast<double*>(smem);
    const int T = a.T, per = R * R + R, r = a.r, perr = r * r + r;   // per: padded image stride; perr: actual entries
    double* tab = lds + ((img_doubles<R>(a.S) + 1) & ~1);
    double* red = tab + QN_TANH_LDS_DOUBLES + 1;                 // 8 doubles
    double* stash = red + 8;                                     // [2 S r][T]
    double* accW = stash + (size_t)2 * a.S * r * T;              // [S * perr][T]
    const int b = blockIdx.y, tid = threadIdx.x;
    stage<R>(a, W + (int64_t)b * a.p, lds, tid, T);
    qn_tanh_table_stage(tab, tid, T);
    for (int e = tid; e < a.S * perr * T; e += T) accW[e] = 0.0;
    __syncthreads();
    const double* Wpre = lds;
    const double* Weff = lds + R * DOMAX + R;
    const double* Wpost = Weff + a.S * per;
    const double* bpost = Wpost + DOMAX * R;
    const double h = 1.0 / a.S, sc = a.mlp ? 1.0 : h;
    double sse = 0.0;
    double gWpre[R][DOMAX], gbpre[R], gWpost[DOMAX][R], gbpost[DOMAX];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        gbpre[j] = 0.0;
#pragma unroll
        for (int k = 0; k < DOMAX; ++k) gWpre[j][k] = 0.0;
    }
#pragma unroll
    for (int q = 0; q < DOMAX; ++q) {
        gbpost[q] = 0.0;
#pragma unroll
        for (int k = 0; k < R; ++k) gWpost[q][k] = 0.0;
    }
    double* mystash = stash + tid;
    double* myacc = accW + tid;
    for (int n = blockIdx.x * T + tid; n < a.Nb; n += gridDim.x * T) {
        const int64_t row = row_idx ? (int64_t)row_idx[(int64_t)b * a.Nb + n] : (int64_t)n;
        double x[DOMAX];
#pragma unroll
        for (int k = 0; k < DOMAX; ++k) x[k] = k < a.d ? X[row * a.d + k] : 0.0;
        double out[R];
        forward_row<R, true>(a, lds, tab, x, out, mystash, T);
        // ---- last layer, residual; g = d SSE / d out_S
        double g[R];
#pragma unroll
        for (int k = 0; k < R; ++k) g[k] = 0.0;
#pragma unroll
        for (int q = 0; q < DOMAX; ++q) {
            if (q >= a.o) break;
            double pr;
            if (a.post) {
                pr = bpost[q];
#pragma unroll
                for (int k = 0; k < R; ++k) pr = fma(Wpost[q * R + k], out[k], pr);
            } else {
                pr = out[q < R ? q : 0];
            }
            const double res = pr - Y[row * a.o + q];
            sse += res * res;
            if (pred) pred[((int64_t)b * a.Nb + n) * a.o + q] = pr;
            const double dl = 2.0 * res;
            if (a.post) {
                gbpost[q] += dl;
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    gWpost[q][k] = fma(dl, out[k], gWpost[q][k]);
                    g[k] = fma(Wpost[q * R + k], dl, g[k]);
                }
            } else {
#pragma unroll
                for (int k = 0; k < R; ++k)
                    if (k == q) g[k] = dl;
            }
        }
        // ---- residual steps, backwards
        for (int i = a.S - 1; i >= 0; --i) {
            const double* Wi = Weff + i * per;
            double oi[R], dz[R];
#pragma unroll
            for (int j = 0; j < R; ++j) {
                oi[j] = j < r ? mystash[(size_t)((2 * i) * r + j) * T] : 0.0;
                const double th = j < r ? mystash[(size_t)((2 * i + 1) * r + j) * T] : 0.0;
                dz[j] = j < r ? sc * g[j] * act_d(th, a.act) : 0.0;
            }
            double* ai = myacc + (size_t)i * perr * T;
            double gn[R];
#pragma unroll
            for (int k = 0; k < R; ++k) gn[k] = a.mlp ? 0.0 : g[k];
#pragma unroll
            for (int j = 0; j < R; ++j) {
                if (j < r) {
                    ai[(size_t)(r * r + j) * T] += dz[j];
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        if (k < r) ai[(size_t)(j * r + k) * T] = fma(dz[j], oi[k], ai[(size_t)(j * r + k) * T]);
                        gn[k] = fma(Wi[j * R + k], dz[j], gn[k]);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < R; ++k) g[k] = gn[k];
        }
        // ---- pre layer: out_0 = act(Wpre x + bpre) is the first stash row
        if (a.pre) {
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const double o0 = j < r ? mystash[(size_t)j * T] : 0.0;
                const double dzp = j < r ? g[j] * act_d(o0, a.act) : 0.0;
                gbpre[j] += dzp;
#pragma unroll
                for (int k = 0; k < DOMAX; ++k) gWpre[j][k] = fma(dzp, x[k], gWpre[j][k]);
            }
        }
    }
    // ---- block reduction -> slab[b][blk][img_doubles]
    __syncthreads();
    const int nimg = img_doubles<R>(a.S);
    double* dst = slab + ((int64_t)b * a.nblk + blockIdx.x) * nimg;
    // step gradients: entry e summed over the T thread-private columns (fixed order)
    for (int e = tid; e < a.S * per; e += T) {                   // e in the padded image layout [S][R*R + R]
        const int i = e / per, q = e % per;
        int src = -1;                                             // row of the compact accumulator [S][r*r + r]
        if (q < R * R) {
            const int j = q / R, k = q % R;
            if (j < r && k < r) src = i * perr + j * r + k;
        } else if (q - R * R < r) {
            src = i * perr + r * r + (q - R * R);
        }
        double s = 0.0;
        if (src >= 0) {
            const double* rowp = accW + (size_t)src * T;
            for (int t = 0; t < T; ++t) s += rowp[(t + e) % T];    // rotate the start: no bank conflict between threads
        }
        dst[R * DOMAX + R + e] = s;
    }
    // pre / post gradients: registers -> wave sums -> LDS scratch (reuse the stash) -> fixed-order sum over waves
    __syncthreads();
    const int nw = T / 64, wave = tid >> 6, lane = tid & 63;
    double* scr = stash;                                          // [nw][R*DOMAX + R + DOMAX*R + DOMAX]
    const int nsm = R * DOMAX + R + DOMAX * R + DOMAX;
#pragma unroll
    for (int j = 0; j < R; ++j) {
#pragma unroll
        for (int k = 0; k < DOMAX; ++k) {
            const double v = wave_sum(gWpre[j][k]);
            if (lane == 0) scr[wave * nsm + j * DOMAX + k] = v;
        }
        const double v = wave_sum(gbpre[j]);
        if (lane == 0) scr[wave * nsm + R * DOMAX + j] = v;
    }
#pragma unroll
    for (int q = 0; q < DOMAX; ++q) {
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const double v = wave_sum(gWpost[q][k]);
            if (lane == 0) scr[wave * nsm + R * DOMAX + R + q * R + k] = v;
        }
        const double v = wave_sum(gbpost[q]);
        if (lane == 0) scr[wave * nsm + R * DOMAX + R + DOMAX * R + q] = v;
    }
    sse = wave_sum(sse);
    if (lane == 0) red[wave] = sse;
    __syncthreads();
    for (int e = tid; e < nsm; e += T) {
        double s = 0.0;
        for (int w = 0; w < nw; ++w) s += scr[w * nsm + e];
        if (e < R * DOMAX + R) dst[e] = s;
        else dst[a.S * per + e] = s;                              // post block sits behind the step gradients
    }
    if (tid == 0) {
        double s = 0.0;
        for (int w = 0; w < nw; ++w) s += red[w];
        partial[(int64_t)b * a.nblk + blockIdx.x] = s;
    }
}

// gradW[b][:] from the per-block partials: sum over blocks (fixed order), contraction of the step gradients
template <int R>
__global__ __launch_bounds__(256) void k_rnet_grad_reduce(RnFusedArgs a, const double* __restrict__ slab,
                                                          double* __restrict__ gradW) {
    const int b = blockIdx.y;
    const int per = R * R + R, nimg = img_doubles<R>(a.S), r = a.r;
    const double* sb = slab + (int64_t)b * a.nblk * nimg;
    double* gb = gradW + (int64_t)b * a.p;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < a.p; e += (int64_t)gridDim.x * blockDim.x) {
        double s = 0.0;
        if (a.pre && e >= a.offWpre && e < a.offBpre) {
            const int q = (int)(e - a.offWpre), j = q / a.d, k = q % a.d;
            for (int m = 0; m < a.nblk; ++m) s += sb[(int64_t)m * nimg + j * DOMAX + k];
        } else if (a.pre && e >= a.offBpre && e < a.offBpre + r) {
            const int j = (int)(e - a.offBpre);
            for (int m = 0; m < a.nblk; ++m) s += sb[(int64_t)m * nimg + R * DOMAX + j];
        } else if (a.post && e >= a.offWpost && e < a.offBpost) {
            const int q = (int)(e - a.offWpost), qo = q / r, k = q % r;
            for (int m = 0; m < a.nblk; ++m) s += sb[(int64_t)m * nimg + R * DOMAX + R + a.S * per + qo * R + k];
        } else if (a.post && e >= a.offBpost && e < a.offBpost + a.o) {
            const int qo = (int)(e - a.offBpost);
            for (int m = 0; m < a.nblk; ++m) s += sb[(int64_t)m * nimg + R * DOMAX + R + a.S * per + DOMAX * R + qo];
        } else if (e >= a.offWW && e < a.offWW + (int64_t)a.npar * r * r) {
            const int q = (int)(e - a.offWW), kpar = q / (r * r), jk = q % (r * r), j = jk / r, k = jk % r;
            for (int i = 0; i < a.S; ++i) {
                double si = 0.0;
                for (int m = 0; m < a.nblk; ++m) si += sb[(int64_t)m * nimg + R * DOMAX + R + i * per + j * R + k];
                if (a.use[i * a.npar + kpar]) s = fma(a.coef[i * a.npar + kpar], si, s);
            }
        } else if (a.has_bias && e >= a.offBB && e < a.offBB + (int64_t)a.npar * r) {
            const int q = (int)(e - a.offBB), kpar = q / r, j = q % r;
            for (int i = 0; i < a.S; ++i) {
                double si = 0.0;
                for (int m = 0; m < a.nblk; ++m) si += sb[(int64_t)m * nimg + R * DOMAX + R + i * per + R * R + j];
                if (a.use[i * a.npar + kpar]) s = fma(a.coef[i * a.npar + kpar], si, s);
            }
        }
        gb[e] = s;
    }
}

__global__ void k_rnet_sse_final(const double* __restrict__ partial, int nblk, int B, double* __restrict__ sse) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double s = 0.0;
    for (int i = 0; i < nblk; ++i) s += partial[(int64_t)b * nblk + i];
    sse[b] = s;
}

int pad_r(int r) { return r <= 4 ? 4 : 8; }

template <int R, int DO = DOMAX> size_t fwd_lds(int S) {
    return sizeof(double) * (size_t)(((img_doubles<R, DO>(S) + 1) & ~1) + QN_TANH_LDS_DOUBLES + 1 + 8);
}
template <int R> size_t bwd_lds(int S, int T, int r) {
    return fwd_lds<R>(S) + sizeof(double) * (size_t)(2 * S * r + S * (r * r + r)) * T;
}
constexpr size_t LDS_BUDGET = 150 * 1024;

template <int R> int pick_T(int S, int r) {
    for (int T : {256, 128, 64})
        if (bwd_lds<R>(S, T, r) <= LDS_BUDGET) return T;
    return 0;
}

void fill_args(const qn_desc* d, int B, int N, int Nb, RnFusedArgs* a) {
    a->p = d->p; a->B = B; a->N = N; a->Nb = Nb; a->d = d->dims[0]; a->o = d->dims[2]; a->r = d->rn_r; a->S = d->rn_steps;
    a->npar = d->rn_npar; a->pre = d->rn_pre; a->post = d->rn_post; a->mlp = d->rn_mlp; a->has_bias = d->has_bias;
    a->act = d->act;
    a->offWpre = d->rn_offWpre; a->offBpre = d->rn_offBpre; a->offWpost = d->rn_offWpost; a->offBpost = d->rn_offBpost;
    a->offWW = d->rn_offWW; a->offBB = d->rn_offBB;
    for (int i = 0; i < d->rn_steps * d->rn_npar; ++i) a->coef[i] = d->rn_coef[i];
    for (int i = 0; i < d->rn_steps * d->rn_npar; ++i) a->use[i] = d->rn_uses[i];
}

int blocks_for(int B, int Nb, int T) {
    int nb = (Nb + T - 1) / T;
    int want = (1024 + B - 1) / B;          // ~4 blocks per CU over the chip
    if (want < 1) want = 1;
    if (nb > want) nb = want;
    if (nb > 64) nb = 64;
    return nb < 1 ? 1 : nb;
}

template <int R>
int run(const qn_desc* d, RnFusedArgs& a, const double* W, const double* X, const double* Y, const int32_t* row_idx,
        double* sse, double* pred, double* gradW, void* ws, size_t ws_bytes, hipStream_t st) {
    const bool grad = gradW != nullptr;
    a.T = grad ? pick_T<R>(a.S, a.r) : 256;
    a.nblk = blocks_for(a.B, a.Nb, a.T);
    const size_t npart = qn_align((size_t)a.B * a.nblk * sizeof(double));
    const size_t nslab = grad ? qn_align((size_t)a.B * a.nblk * img_doubles<R>(a.S) * sizeof(double)) : 0;
    if (npart + nslab > ws_bytes) {
        qn_set_error("workspace too small: need %zu bytes, got %zu", npart + nslab, ws_bytes);
        return QN_EWORKSPACE;
    }
    double* partial = static_cast<double*>(ws);
    double* slab = reinterpret_cast<double*>(static_cast<char*>(ws) + npart);
    (void)hipGetLastError();
    if (!grad && (a.d > DOMAX || a.o > DOMAX)) {
        static bool armed = false;
        if (!armed) {
            QN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rnet_fwd<R, DOWIDE>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            armed = true;
        }
        const size_t lds_w = fwd_lds<R, DOWIDE>(a.S);
        hipLaunchKernelGGL((k_rnet_fwd<R, DOWIDE>), dim3(a.nblk, a.B), dim3(256), lds_w, st, a, W, X, Y, row_idx, pred,
                           partial);
    } else if (!grad) {
        static bool armed = false;
        if (!armed) {
            QN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rnet_fwd<R>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            armed = true;
        }
        hipLaunchKernelGGL(k_rnet_fwd<R>, dim3(a.nblk, a.B), dim3(256), fwd_lds<R>(a.S), st, a, W, X, Y, row_idx, pred,
                           partial);
    } else {
        static bool armed = false;
        if (!armed) {
            QN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rnet_bwd<R>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            armed = true;
        }
        hipLaunchKernelGGL(k_rnet_bwd<R>, dim3(a.nblk, a.B), dim3(a.T), bwd_lds<R>(a.S, a.T, a.r), st, a, W, X, Y, row_idx, pred,
                           partial, slab);
        int gx = (int)((a.p + 255) / 256);
        if (gx > 16) gx = 16;
        hipLaunchKernelGGL(k_rnet_grad_reduce<R>, dim3(gx, a.B), dim3(256), 0, st, a, (const double*)slab, gradW);
    }
    hipLaunchKernelGGL(k_rnet_sse_final, dim3((a.B + 63) / 64), dim3(64), 0, st, (const double*)partial, a.nblk, a.B, sse);
    QN_HIP_CHECK(hipGetLastError());
    return QN_OK;
}

}  // namespace

bool qn_rnet_fused_supported(const qn_desc* d, int want_grad, int dtype) {
    if (d->kind != QN_KIND_RNET || dtype != QN_F64) return false;
    const int dmax = want_grad ? DOMAX : DOWIDE;
    if (d->rn_r > RMAX || d->dims[0] > dmax || d->dims[2] > dmax) return false;
    if (d->act != QN_ACT_TANH && d->act != QN_ACT_IDENTITY) return false;
    if (!want_grad) return true;
    return (pad_r(d->rn_r) == 4 ? pick_T<4>(d->rn_steps, d->rn_r) : pick_T<8>(d->rn_steps, d->rn_r)) > 0;
}

size_t qn_rnet_fused_workspace(const qn_desc* d, int B, int Nb, int want_grad) {
    const int R = pad_r(d->rn_r);
    const int T = want_grad ? (R == 4 ? pick_T<4>(d->rn_steps, d->rn_r) : pick_T<8>(d->rn_steps, d->rn_r)) : 256;
    const int nblk = blocks_for(B, Nb, T ? T : 256);
    const int nimg = R == 4 ? img_doubles<4>(d->rn_steps) : img_doubles<8>(d->rn_steps);
    return qn_align((size_t)B * nblk * sizeof(double)) + (want_grad ? qn_align((size_t)B * nblk * nimg * sizeof(double)) : 0) + 256;
}

int qn_rnet_fused_run(const qn_desc* d, const void* W, const void* X, const void* Y, const int32_t* row_idx, int B, int N,
                      int Nb, double* sse, void* pred, void* gradW, void* ws, size_t ws_bytes, hipStream_t st) {
    RnFusedArgs a;
    fill_args(d, B, N, Nb, &a);
    if (pad_r(d->rn_r) == 4)
        return run<4>(d, a, (const double*)W, (const double*)X, (const double*)Y, row_idx, sse, (double*)pred, (double*)gradW,
                      ws, ws_bytes, st);
    return run<8>(d, a, (const double*)W, (const double*)X, (const double*)Y, row_idx, sse, (double*)pred, (double*)gradW, ws,
                  ws_bytes, st);
}
