/* quinn_amd.h -- C ABI of the MI355X (gfx950) hot path for QUiNN.
 *
 * The reference (sandialabs/quinn) is pure Python and has no FFI layer; its operator
 * boundaries for this path are Python callables.  Each entry point below replaces the
 * inner loop behind one of them and is what a ctypes binding in the reference would call
 * (INTEGRATION.md shows the stubs).  Citations are file:line relative to the reference.
 *
 * Conventions
 *   - plain pointers and sizes only; every data pointer is a BORROWED DEVICE pointer
 *     (HIP), never allocated or freed across this boundary;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all work is
 *     enqueued on it, no call synchronises the device;
 *   - return value: QN_OK (0) or a negative QN_E* code; qn_last_error() gives the text
 *     of the calling thread's last failure;
 *   - dtype: QN_F64 (the reference's arithmetic, quinn/nns/tchutils.py:9) or QN_F32.
 *
 * Flat parameter layout of one weight vector (quinn/nns/nnwrap.py:70-77, i.e.
 * module.parameters() order of quinn/nns/mlp.py:55-84):
 *   [ W_0 (h_1 x d, row-major), b_0 (h_1), W_1 (h_2 x h_1), b_1, ..., W_L (o x h_L), b_L ]
 */
#ifndef QUINN_AMD_H
#define QUINN_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { QN_F64 = 0, QN_F32 = 1 };
enum { QN_ACT_IDENTITY = 0, QN_ACT_TANH = 1, QN_ACT_RELU = 2 };
enum { QN_OK = 0, QN_EINVAL = -1, QN_EWORKSPACE = -2, QN_EHIP = -3, QN_EUNSUPPORTED = -4 };
/* kernel families, for qn_mlp_desc_set_path (tests / profiling) */
enum { QN_PATH_AUTO = 0, QN_PATH_GENERIC = 1, QN_PATH_FUSED = 2, QN_PATH_FUSED_DP = 3 };

typedef struct qn_desc qn_desc;

/* Architecture descriptor of a quinn-style MLP: dims = {d, h_1, ..., h_L, o} (ndims >= 2),
 * one activation between consecutive Linear layers, none after the last
 * (quinn/nns/mlp.py:46-84: 'tanh' | 'relu' | identity). */
int qn_mlp_desc_create(const int* dims, int ndims, int act, int has_bias, qn_desc** out);
/* Descriptor of a quinn residual network (quinn/nns/rnet.py:16-165):
 *   out = act(Wpre x + bpre)                       if layer_pre, else out = x (indim == rdim)
 *   for i in 0..nsteps-1:  W_i = sum_k coef[i*npar+k] * ww_k,  b_i likewise from bb_k (if has_bias)
 *       out = mlp ? act(W_i out + b_i) : out + (1/nsteps) * act(W_i out + b_i)
 *   pred = Wpost out + bpost                       if layer_post, else pred = out (outdim == rdim)
 * nsteps = nlayers + 1 of the reference; coef[i][k] is the weight-parameterisation function
 * (rnet.py:217-380) evaluated at t = i / nsteps: t^k for Const/Lin/Quad/Cubic/Poly, a one-hot at
 * int(t * npar) for NonPar.  act = tanh (nonlin=True) or identity.  Flat layout = the module's
 * parameters() order: weight_pre, bias_pre, weight_post, bias_post, ww_0.., bb_0.. (rnet.py:90-120).
 * The descriptor is used with the same qn_mlp_* entry points as an MLP's. */
int qn_rnet_desc_create(int indim, int rdim, int outdim, int nsteps, int npar, const double* coef, int act,
                        int has_bias, int layer_pre, int layer_post, int mlp, qn_desc** out);
/* uses[i*npar+k] = 0: parameter tensor k does not enter step i at all (NonPar, rnet.py:349-377, picks ONE tensor per step:
 * `pars[int(t * npar)]`); such a tensor is skipped -- its value, even Inf or NaN, has no effect on the step and its
 * gradient receives nothing from it -- where a tensor with uses = 1 and coefficient 0 (the t^k of Lin / Quad / Cubic / Poly at
 * t = 0, which the reference multiplies out) contributes 0 * value.  Default: every entry 1.  n = nsteps * npar; an entry
 * marked unused must have coefficient 0. */
int qn_rnet_desc_set_uses(qn_desc* desc, const unsigned char* uses, int n);
int qn_mlp_desc_destroy(qn_desc* desc);
/* p = number of entries of one flat weight vector. */
int64_t qn_mlp_num_params(const qn_desc* desc);

/* Bytes of scratch the two calls below need for B weight vectors x Nb rows each.  The workspace is the caller's: it needs
 * no initialisation and keeps no state between calls (the fused forward kernels sum a chain's SSE by a last-arriver
 * protocol on a tagged word of it that treats any other content as "fresh" and resets itself), but it must not be shared by
 * calls that can run concurrently (two streams: two workspaces). */
size_t qn_workspace_bytes(const qn_desc* desc, int B, int Nb, int want_grad, int dtype);

/* Which kernel family the next call with these sizes would run (QN_PATH_GENERIC/FUSED). */
int qn_mlp_path(const qn_desc* desc, int B, int Nb, int want_grad, int dtype);
/* Which ARITHMETIC the next call with these sizes would form the hidden-layer products in: QN_ARITH_PLAIN -- the arithmetic
 * of `dtype` throughout; QN_ARITH_I8_FUSED -- sliced exact int8 products in the one-launch kernels of 64-wide networks
 * (qn_fused_i8.hip / qn_fused_bwd_i8.hip; also a narrower network's zero-padded 64-wide twin); QN_ARITH_I8_WIDE -- the
 * int8-slice kernels of 128 / 256-wide networks (qn_wide_i8.hip; tanh also qn_dw_i8.hip); QN_ARITH_I8_LAYERS -- the layer-wise
 * int8-slice forward of other tanh networks whose widths are multiples of 64.  (The reference has one arithmetic, torch
 * float64: quinn/nns/tchutils.py:9; tests use this query to prove which kernels a parity case exercised.) */
enum { QN_ARITH_PLAIN = 0, QN_ARITH_I8_FUSED = 1, QN_ARITH_I8_WIDE = 2, QN_ARITH_I8_LAYERS = 3 };
int qn_mlp_arith(const qn_desc* desc, int B, int Nb, int want_grad, int dtype);
/* Force a kernel family for the calls made with THIS descriptor (QN_PATH_AUTO restores dispatch by shape).  Returns
 * the previous setting.  There is no process-wide state: two operators in one process do not see each other's
 * choice.  QN_PATH_GENERIC is the layer-wise family at the EXACT layer widths; under QN_PATH_AUTO hidden widths that
 * are no multiples of 64 run on a zero-padded twin of the network (padded units stay exactly 0, results equal the
 * unpadded network's).  QN_PATH_FUSED_DP is QN_PATH_FUSED restricted to the kernels that use the float64 matrix
 * instructions: it excludes the forward kernel that forms the 64-wide hidden layers as sliced exact int8 products
 * (same results to ~1e-14 relative; kept selectable as the second implementation the tests compare it with).  Under
 * QN_PATH_AUTO the float64 networks with hidden widths all 128 or all 256 (one output, <= 8 inputs) take the layer-wise
 * family with their hidden layers -- forward, activation gradient and, for tanh, weight gradient -- as sliced exact int8
 * products (~1e-13 relative; relu / identity: one activation scale per data row and layer, and every weight and bias below
 * 2^20, inputs below 2^100); QN_PATH_GENERIC is the all-float64 reference of that family as well.
 * Shapes of the one-launch (QN_PATH_FUSED) kernels: uniform hidden width 16 / 32 / 64 (any width <= 64 through the twin), up to
 * 4 (64-wide: 3) hidden layers in a gradient call; forward and gradient: up to 16 inputs and 16 outputs (gradient of three
 * 64-wide hidden layers: not more than 4 inputs together with more than 4 outputs).  Everything else runs layer-wise.
 * Accuracy of the int8-slice kernels: operands are rounded to 2^-47 of (1 x the weight row's maximum), the products are
 * exact -- a norm-wise bound, 47-bit against float64's 53; rows whose activations are all below ~2^-5 and chains with a
 * hidden-matrix weight >= 2^20 (or not finite) are computed in plain float64 instead.
 * Not-finite values, every family: SSE, predictions and gradient carry the NaN / +Inf / -Inf pattern of the reference's
 * torch ops on the same inputs (relu(NaN) = NaN, relu backward is a select, tanh saturates an infinite input); a gradient
 * entry that is +-Inf there may be NaN here.  Chains of a zero-padded twin with an unbounded weight or input are recomputed
 * from the original weights (slow). */
int qn_mlp_desc_set_path(qn_desc* desc, int path);

/* The fused kernels give every chain ceil(512 / B) (gradient: 256 / B) workgroups, each summing its share of the data rows;
 * the per-chain SSE / gradient adds those shares left to right, so its last bits depend on B.  batch > 0: split as a launch
 * of max(B, batch) chains would -- a set of chains run as several smaller launches (the chain groups of the device samplers,
 * quinn_amd/mcmc/device_amcmc.py: one group's accept kernel overlaps the other's forward) then reproduces the one-launch
 * results bit for bit.  0 (default): by the launch's own B.  Returns the previous value. */
int qn_mlp_desc_set_plan_batch(qn_desc* desc, int batch);

/* sse_out[b] = sum_{n,o} (Y[r(b,n),o] - f_{W[b]}(X[r(b,n),:])[o])^2 for b < B.
 * Replaces the per-weight-vector loop over NN_MCMC.logpost -> NNWrap.calc_loss ->
 * NegLogPost.forward -> MLP.forward (quinn/solvers/nn_mcmc.py:45-71,
 * quinn/nns/nnwrap.py:109-126, quinn/nns/losses.py:197-198, quinn/nns/mlp.py:92-101), the
 * per-sample loop of BNet.sample_elbo (quinn/vi/bnet.py:202-205) and the forward of one
 * ensemble member's loss in nnfit (quinn/nns/nnfit.py:133-140); with pred_out also
 * nn_p / Learner.predict (quinn/nns/nnwrap.py:330-347, quinn/ens/learner.py:75-93).
 *   W        [B, p]   dtype, row-major flat weight vectors (layout above)
 *   X        [N, d]   dtype, row-major shared dataset
 *   Y        [N, o]   dtype
 *   row_idx  [B, Nb]  int32 or NULL. NULL: Nb must equal N and r(b,n) = n (all vectors see
 *                     the whole dataset); else r(b,n) = row_idx[b*Nb+n] (per-member
 *                     minibatch / data subset, quinn/nns/nnfit.py:131, nn_ens.py:63-64)
 *   sse_out  [B]      float64 always
 *   pred_out [B, Nb, o] dtype or NULL
 * The scalar tails (log-posterior, NLL, MSE) are applied by the caller in float64. */
int qn_mlp_sse_fwd(const qn_desc* desc, int dtype, const void* W, const void* X, const void* Y,
                   const int32_t* row_idx, int B, int N, int Nb, double* sse_out, void* pred_out,
                   void* workspace, size_t workspace_bytes, void* stream);

/* The forward call without its final summation kernel, for a consumer that sums a handful of numbers itself
 * (qn_mcmc_accept): sse_parts_out is [B, parts], parts = qn_mlp_sse_parts(desc, B, Nb, dtype) >= 1, and the plain
 * left-to-right sum of row b is exactly (bit for bit) what qn_mlp_sse_fwd writes to sse_out[b].  parts > 1 only where
 * the fused forward kernel runs unpadded (one partial per row split of a chain); otherwise parts = 1 and the call is
 * qn_mlp_sse_fwd without predictions.  (qn_mlp_sse_fwd itself sums the parts inside the forward kernel since round 3 -- the
 * last workgroup of a chain to finish, ~1.5 us; this entry point saves that as well.) */
int qn_mlp_sse_parts(const qn_desc* desc, int B, int Nb, int dtype);
int qn_mlp_sse_fwd_parts(const qn_desc* desc, int dtype, const void* W, const void* X, const void* Y,
                         const int32_t* row_idx, int B, int N, int Nb, double* sse_parts_out, void* workspace,
                         size_t workspace_bytes, void* stream);

/* As above plus gradW_out[b, :] = d sse_out[b] / d W[b, :]  ([B, p] dtype).  Replaces
 * NN_MCMC.logpostgrad -> NNWrap.calc_lossgrad (quinn/solvers/nn_mcmc.py:73-98,
 * quinn/nns/nnwrap.py:128-150) and loss.backward() in nnfit (quinn/nns/nnfit.py:163-165). */
int qn_mlp_sse_fwdbwd(const qn_desc* desc, int dtype, const void* W, const void* X, const void* Y,
                      const int32_t* row_idx, int B, int N, int Nb, double* sse_out, void* pred_out,
                      void* gradW_out, void* workspace, size_t workspace_bytes, void* stream);

/* Mean-field Gaussian VI, sampling + KL terms (quinn/vi/bnet.py:142-163,
 * quinn/rvar/rvs.py:96-127, 159-173; sigma = exp(rho), bnet.py:80):
 *   W_out[s,i]  = mu[i] + exp(rho[i]) * eps[s,i]                          [S, p] dtype
 *   logq_out[s] = sum_i( -log sqrt(2 pi) - rho[i] - (w-mu)^2 / (2 sigma^2) )   float64
 *   logp_out[s] = sum_i log( pi N(w;0,sigma1) + (1-pi) N(w;0,sigma2) )          float64
 * mu, rho: [p] float64 (the variational parameters stay float64); eps: [S, p] float64. */
int qn_vi_sample_kl(const double* mu, const double* rho, const double* eps, int S, int64_t p,
                    double pi, double sigma1, double sigma2, int dtype, void* W_out,
                    double* logq_out, double* logp_out, void* stream);

/* Chain rule of viloss = (mean_s logq - mean_s logp) * kl_scale + NLL  (bnet.py:229-232)
 * to (mu, rho), given gW[s,:] = d NLL / d W[s,:] * gw_scale taken from qn_mlp_sse_fwdbwd
 * (gw_scale = 0.5 / (S * o * sigma_d^2) turns dSSE into dNLL, bnet.py:215):
 *   dmu[i]  = sum_s gW[s,i]*gw_scale - kl_scale/S * sum_s gp(w_si)
 *   drho[i] = sum_s gW[s,i]*gw_scale*sigma_i*eps_si - kl_scale*(1 + 1/S sum_s gp(w_si)*sigma_i*eps_si)
 * with gp = d/dw log prior density.  dmu, drho: [p] float64. */
int qn_vi_grad(const double* mu, const double* rho, const double* eps, const void* gW, int S,
               int64_t p, double pi, double sigma1, double sigma2, double gw_scale, double kl_scale,
               int dtype, double* dmu_out, double* drho_out, void* stream);

/* One Adam step for B independent members at once (torch.optim.Adam defaults as used by
 * quinn/nns/nnfit.py:74-75, single-tensor update order):
 *   g = G*gscale + wd*W;  m += (g-m)*(1-b1);  v = v*b2 + (1-b2)*g*g;
 *   W -= lr[b]/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 * W, m, v: [B, p] float64 master state; G: [B, p] dtype; lr: [B] float64 (device);
 * step t >= 1.  Members with lr[b] == 0 are left untouched. */
int qn_adam_batched(double* W, const void* G, double* m, double* v, const double* lr, int B,
                    int64_t p, int dtype, double gscale, double wd, double beta1, double beta2,
                    double eps, int step, void* stream);

/* Device-resident Metropolis-Hastings step (throughput engine of the adaptive Metropolis sampler,
 * quinn/mcmc/admcmc.py:38-74 + quinn/mcmc/mcmc.py:65-85), two kernels around the batched
 * log-posterior.  The step counter lives in device memory (keeping it there makes a run of steps a static
 * launch sequence / HIP graph).  It is DOUBLE-BUFFERED by the parity of the step, like the other per-chain
 * scalars qn_mcmc_accept maintains: the accept call of a step reads slot `parity` and writes slot 1 - parity, so
 * its workgroups (several per chain) never see a half-updated state and need no fence or atomic.  The proposal
 * kernels take a pointer to the CURRENT slot (`step_ptr` of qn_mcmc_propose* / qn_mcmc_apply_delta = base + parity).
 *
 * qn_mcmc_propose: out[c,:] = cur[c,:] + sd[c,:] * z + c1 * z0_c with z ~ N(0,I), z0_c ~ N(0,1)
 *   (the initial proposal covariance c1^2 + diag(sd^2) = 0.01 + diag(0.09|x0|), admcmc.py:65);
 *   with cur == NULL it writes the standard normals z themselves (input of a dense factor product).
 *   Random numbers: Philox4x32-10 keyed by (seed, step, GLOBAL chain id = chain0 + c, purpose, index): a
 *   chain's draws do not depend on how the chains are split over launches or ranks (every qn_mcmc_* call
 *   takes chain0, the global id of its first chain). */
int qn_mcmc_propose(const double* cur, const double* sd, double c1, int C, int chain0, int64_t p, uint64_t seed,
                    const int64_t* step_ptr, double* out, void* stream);

/* qn_mcmc_accept: for every chain c: log-posterior of the proposal from its SSE (sse_prop [C, nparts]: summed left to
 *   right, nparts = 1 for a plain [C] vector; see qn_mlp_sse_fwd_parts),
 *   lp = -(0.5 sse/sigma^2 + (n_rows/2) log 2pi + n_rows log sigma); mh = exp(lp - cur_lp[c]);
 *   accept iff u_c < mh (mcmc.py:72-75); updates cur, cur_lp, best / best_lp (MAP, mcmc.py:79-81),
 *   nacc, writes chain[c, step+1, :] (optional), lps[c, step+1], alphas[c, step+1]; then advances the
 *   counter.  cur_lp, best_lp, kcur: [2, C] and step_ptr: [2] -- slot `parity` (0 / 1) is read, slot 1 - parity
 *   written; the caller alternates parity from step to step (slot 0 holds the initial state for parity 0).
 *   With hist != NULL it also maintains what the adapted proposal is drawn from
 *   (qn_mcmc_propose_hist): hist [C, kcap, pstride] FLOAT16 = the DISTINCT states visited, as (x - x0[c]) * hscale[c]
 *   (x0 [C, p] float64: the chain's reference point -- the start, or wherever the caller re-bases the rows to; hscale [C]
 *   float64 or NULL = 1: a power of two that keeps the rows in float16's range, values are clamped to +-65504; half the bytes
 *   of float32 rows for the kernel that streams the whole history, and its products on the float16 matrix cores; rounding
 *   2^-11 of |x - x0|) (row 0 = the start, a new row per accepted move), mult [C, kcap] their multiplicities in the
 *   chain so far, kcur [2, C] the index of the current state's row, sumx [C, p] the sum of mult x (x - x0) over the
 *   states the chain has LEFT (a stay is added when the chain leaves the state -- no per-step pass over the vector; the sum
 *   of (x_i - x0) over all samples so far is sumx + mult[c, kcur[c]] x (cur[c] - x0[c])).  kcur[c] >= kcap means the history
 *   is full: rows, multiplicities and the sum are no longer maintained (the caller compresses before that happens). */
int qn_mcmc_accept(const double* prop, const double* sse_prop, double sigma, int n_rows, int C, int chain0, int64_t p,
                   int nmcmc, uint64_t seed, double* cur, double* cur_lp, double* best, double* best_lp,
                   double* chain, double* lps, double* alphas, int64_t* nacc, const double* x0, void* hist,
                   const double* hscale, int32_t* mult, int32_t* kcur, double* sumx, int kcap, int64_t pstride,
                   int64_t* step_ptr, int parity, int nparts, void* stream);

/* qn_mcmc_accept_propose: qn_mcmc_accept that also writes the NEXT step's proposal from the state it has just decided
 * (one launch and one pass over the state fewer per step): next_mode 0 = nothing more; 1 = prop_next = cur' + sd z +
 * c1 z0 (qn_mcmc_propose); 2 = prop_next = cur' + delta[c, t_next, :] + s_iso z (qn_mcmc_apply_delta).  Same random
 * numbers (streams of step + 1) and the same arithmetic as the separate kernels: results are bit-identical.
 * prop_next may be the buffer `prop`. */
int qn_mcmc_accept_propose(const double* prop, const double* sse_prop, double sigma, int n_rows, int C, int chain0,
                           int64_t p, int nmcmc, uint64_t seed, double* cur, double* cur_lp, double* best, double* best_lp,
                           double* chain, double* lps, double* alphas, int64_t* nacc, const double* x0, void* hist,
                           const double* hscale, int32_t* mult, int32_t* kcur, double* sumx, int kcap, int64_t pstride,
                           int64_t* step_ptr, int next_mode, const double* sd, double c1, const double* delta, int t_next, double s_iso,
                           double* prop_next, int parity, int nparts, void* stream);

/* qn_mcmc_propose_hist: the ADAPTED proposal of adaptive Metropolis (admcmc.py:52-70), drawn in sample
 *   space.  After an adaptation at step i the reference proposes from N(x, c (cov_i + 1e-8 I)) with
 *   cov_i the unbiased sample covariance of x_0..x_i and c = gamma 2.4^2 / p.  With the K distinct
 *   states x_k of that history, multiplicities w_k, mean m and n = i + 1,
 *       out[c,:] = cur[c,:] + s_lr * sum_k wsnap[c,k] u_k (hist[c,k,:] / hscale[c] - msnap[c,:]) + s_iso * v,
 *   u_k, v_j iid N(0,1), wsnap = sqrt(w), s_lr = sqrt(c/(n-1)), s_iso = sqrt(c 1e-8), has exactly that
 *   covariance: a K x p GEMV over the stored states instead of a p x p factor (cfg2: K ~ 10^2..10^3
 *   rows of 34 KB per chain and step instead of 290-580 MB).  ksnap [C] = K per chain, msnap [C, p] =
 *   mean of (x - x0) at the adaptation, both frozen until the next adaptation.  hist: float16 rows (x_k - x0) * hscale
 *   (qn_mcmc_accept); the coefficients wsnap u_k are rounded to float16 as well (the matrix cores' operand type; both the
 *   single-step and the block kernel use the rounded values, products accumulate in float32 / float64).  pstride even, >= p
 *   (a multiple of 4 for qn_mcmc_propose_hist_block). */
int qn_mcmc_propose_hist(const double* cur, const void* hist, const float* wsnap, const int32_t* ksnap,
                         const double* msnap, const double* hscale, double s_lr, double s_iso, int C, int chain0, int64_t p,
                         int64_t pstride, int kcap, uint64_t seed, const int64_t* step_ptr, double* out, void* stream);

/* The same draw for TB = qn_mcmc_hist_block_steps() (= 64) consecutive steps in one pass over the stored states:
 * the increment of step t depends only on the frozen snapshot and on that step's random numbers (keyed
 * by the absolute step, exactly as in qn_mcmc_propose_hist), not on the chain's state, so
 *   delta[c, t, :] = s_lr * sum_k wsnap[c,k] u_k^(step0+t) (hist[c,k,:] / hscale[c] - msnap[c,:])
 * for t = 0..TB-1 reads the history once: HBM traffic per step / TB, a (TB x K).(K x p) product per chain
 * (v_mfma_f32_32x32x16_f16: float16 operands, float32 accumulation).  step_ptr != NULL: step0 is read from the device step
 * counter when the kernels run (a static launch, capturable in a HIP graph).  coef: scratch of
 * qn_mcmc_hist_block_coef_bytes(C, kcap) bytes; delta: [C, TB, p] float64.
 * order (optional, [C] int32): a permutation of the chains giving the dispatch order of the history product -- pass the
 * chains sorted by ksnap, longest first (the work per chain is proportional to ksnap[c]); results do not depend on it.
 * qn_mcmc_apply_delta: out[c,:] = cur[c,:] + delta[c, t, :] + s_iso * v (the proposal of step step0 + t; v on the
 * stream of the CURRENT step *step_ptr, as in qn_mcmc_propose_hist; s_iso is unused by the block call). */
int qn_mcmc_hist_block_steps(void);
size_t qn_mcmc_hist_block_coef_bytes(int C, int kcap);
int qn_mcmc_propose_hist_block(const void* hist, const float* wsnap, const int32_t* ksnap, const double* msnap,
                               const double* hscale, double s_lr, double s_iso, int C, int chain0, int64_t p, int64_t pstride,
                               int kcap, uint64_t seed, int64_t step0, const int64_t* step_ptr, void* coef, double* delta,
                               const int32_t* order, void* stream);
int qn_mcmc_apply_delta(const double* cur, const double* delta, int t, double s_iso, int C, int chain0, int64_t p,
                        uint64_t seed, const int64_t* step_ptr, double* out, void* stream);

/* Device-resident Hamiltonian Monte Carlo step (throughput engine of quinn/mcmc/hmc.py:43-66 + the accept block of
 * quinn/mcmc/mcmc.py:65-85): three elementwise kernels around qn_mlp_sse_fwdbwd, no host synchronisation, every launch
 * static given the step parity (a pair of steps is capturable as one HIP graph).  All state is float64 [C, p]; the
 * gradient arrays hold d SSE / d W as qn_mlp_sse_fwdbwd writes it (d logpost = -0.5/sigma^2 * d SSE is applied here).
 *
 * qn_hmc_begin (hmc.py:43-51): momentum z ~ N(0, I) -- Philox4x32-10 keyed by (seed, step, GLOBAL chain id = chain0 + c,
 *   index), so a chain's draws do not depend on how chains are split over launches or ranks --,
 *   mom = z + (eps/2) d logpost(cur), q = cur + eps * mom, and the current kinetic energy as partial sums of z^2:
 *   kin_cur_parts [C, qn_hmc_parts(p)] (the accept call adds them left to right and halves the sum; the number of
 *   partials depends on p alone).  grad_cur is the gradient at the current state, kept from the accepted proposal.
 * qn_hmc_leap (hmc.py:53-60): given grad_q = d SSE / d W at q (dtype QN_F64 / QN_F32): inner step (last = 0)
 *   mom += eps * d logpost(q), q += eps * mom; last step (last = 1) mom += (eps/2) * d logpost(q) and kin_prop_parts
 *   [C, qn_hmc_parts(p)] = partial sums of mom^2 (the sign flip of hmc.py:64 does not change it).
 * qn_hmc_accept (mcmc.py:68-85): log-posterior of the proposal from sse_q [C] (the SSE the last gradient call returned:
 *   the reference spends an extra forward pass on it), mh = exp((U + K) - (U' + K')), accept iff u_c < mh; on accept
 *   cur <- q, grad_cur <- grad_q; MAP, chain / lps / alphas rows, nacc and the double-buffered scalars / step counter
 *   exactly as qn_mcmc_accept (cur_lp, best_lp: [2, C]; step_ptr: [2]; slot `parity` read, slot 1 - parity written). */
int qn_hmc_parts(int64_t p);
int qn_hmc_begin(const double* cur, const double* grad_cur, double sigma, double epsilon, int C, int chain0, int64_t p,
                 uint64_t seed, const int64_t* step_ptr, double* mom, double* q, double* kin_cur_parts, void* stream);
int qn_hmc_leap(const void* grad_q, int dtype, double sigma, double epsilon, int last, int C, int64_t p, double* mom,
                double* q, double* kin_prop_parts, void* stream);
int qn_hmc_accept(const double* q, const double* grad_q, const double* sse_q, const double* kin_cur_parts,
                  const double* kin_prop_parts, double sigma, int n_rows, int C, int chain0, int64_t p, int nmcmc,
                  uint64_t seed, double* cur, double* grad_cur, double* cur_lp, double* best, double* best_lp,
                  double* chain, double* lps, double* alphas, int64_t* nacc, int64_t* step_ptr, int parity, void* stream);

/* Moments of a predictive ensemble on the device (QUiNNBase.predict_mom_sample, quinn/solvers/quinn.py:75-104):
 * Y [M, K] dtype = M members x K = N * o prediction entries as qn_mlp_sse_fwd writes them ([B, Nb, o] with B = M);
 * mean_out [K], var_out [K] (unbiased, ddof = 1; NULL to skip; needs M >= 2), float64.  The per-output covariance of
 * msc = 2 is a plain GEMM of the centred ensemble and is left to the BLAS of the host side. */
int qn_pred_moments(const void* Y, int dtype, int64_t M, int64_t K, double* mean_out, double* var_out, void* stream);

/* Diagnostic: y[i] = device tanh(x[i]) in float64 (the activation used by every kernel). */
int qn_debug_tanh(const double* x, double* y, int64_t n, void* stream);
/* Diagnostic: the variant the fused kernels use when all weights and inputs are finite and bounded
 * (no NaN handling; same values for every non-NaN input). */
int qn_debug_tanh_finite(const double* x, double* y, int64_t n, void* stream);
/* Diagnostic: the table-assisted tanh of the fused kernels (tanh(n/16) from LDS + a short polynomial;
 * nansafe = 1: NaN-propagating variant, 0: the variant for arguments that cannot be NaN; 2: the absolute-accuracy variant
 * of the int8-slice forward kernel (tanh(n/64) table, arguments that cannot be NaN). */
int qn_debug_tanh_table(const double* x, double* y, int64_t n, int nansafe, void* stream);

const char* qn_last_error(void);
/* "quinn_amd <version> gfx950" */
const char* qn_version(void);

#ifdef __cplusplus
}
#endif
#endif /* QUINN_AMD_H */
