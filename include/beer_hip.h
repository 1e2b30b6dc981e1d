/*
 * beer_hip.h -- C ABI of the MI355X (gfx950) implementation of beer's
 * variational-Bayes hot path.
 *
 * The reference (beer-asr/beer) is pure Python on torch and has no FFI of its
 * own; the seam is its Python `Model` protocol.  Every
 * entry point below names the reference function it replaces (paths relative
 * to the reference root).  The Python host layer `beer_amd` binds these with
 * ctypes (beer_amd/_hip.py); INTEGRATION.md shows the stub a maintainer of
 * the reference would add to call them from beer itself.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (torch is only
 *     the allocator) unless the name ends in `_h` (host);
 *   - `dtype`: BEER_F32 / BEER_F64 is the element type of every `void*`
 *     floating-point buffer of the call; `double*` buffers are always fp64
 *     (cross-frame accumulators are fp64 whatever the model dtype is);
 *   - matrices are row-major and dense, no padding;
 *   - `stream` is a hipStream_t (0 = default stream); calls are asynchronous,
 *     never synchronise, never allocate device memory, keep no state between
 *     calls and are re-entrant per stream;
 *   - return value: 0 on success, BEER_EINVAL for a bad argument, or
 *     -(hipError_t) if a launch failed.  Nothing throws.
 *   - `cov`: BEER_FULL / BEER_DIAG / BEER_ISO selects the layout of the
 *     sufficient statistics, Q = D*D+D+2 | 2D+2 | D+3
 *     (beer/dists/normalwishart.py:30-38, normalgamma.py:20-27,
 *     isonormalgamma.py:21-30).
 */
#ifndef BEER_HIP_H
#define BEER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BEER_F32 0
#define BEER_F64 1
#define BEER_I16 2   /* audio samples only (beer_features_*) */

#define BEER_FULL 0
#define BEER_DIAG 1
#define BEER_ISO 2

#define BEER_OK 0
#define BEER_EINVAL (-100000)

/* Library / device probes (no reference counterpart). */
int beer_hip_version(void);
int beer_hip_device_count(void);

/* Tuning options (no reference counterpart): process-wide launch parameters of the
 * matrix-core kernels, for measurements and for the tests that pin the defaults.
 * They change how the work is cut into launches and chains, never what is computed;
 * the defaults are what the parity suite validates.  A value outside the range is
 * refused with BEER_EINVAL and leaves the option as it was. */
#define BEER_OPT_AX_MAXFRAMES 0  /* frames a workgroup of beer_normal_accumulate_packed sums in
                                  * float32 before its partial sums meet in fp64 (the matrix
                                  * core truncates its accumulator: bias ~ 8e-11 per frame of
                                  * chain on same-sign sums).  Default 4096, range 64 .. 2^20. */
#define BEER_OPT_ACCF_ROUNDS 1   /* workgroup rounds per CU of beer_mixtureset_accumulate_fused.
                                  * Default 6, range 1 .. 64. */
#define BEER_OPT_K1_WIDE 2       /* 1: the packed E-step runs its 64 x 256-per-wave form.
                                  * Default 0. */
#define BEER_OPT_ACCFI_WAVES 3   /* waves per workgroup of the fused accumulation over a frame
                                  * image: 4 (two workgroups per CU: one flushes its partial
                                  * sums while the other multiplies) or 8.  Default 4. */
#define BEER_OPT_LNFI 4          /* 1: beer_mixtureset_lognorm_image keeps a chunk's packed
                                  * parameters in LDS and walks blocks of frames (lnfi_kernel);
                                  * 0: one tile per wave, parameters streamed from L2.  Default 1. */
#define BEER_OPT_FB_LOG 5        /* 1: the one-wave-per-utterance forward-backward runs EVERY
                                  * utterance in log space (the kernel that otherwise only
                                  * redoes utterances whose dynamic range exceeds the
                                  * scaled-probability recursion; beer/graph.py:270-326 is
                                  * log-space throughout).  Default 0. */
#define BEER_OPT_K1_LDS 6        /* 1: the packed full-covariance E-step stages the packed
                                  * parameters of a k-step through LDS once per workgroup (DMA,
                                  * ring of two half k-steps) instead of streaming them from L2
                                  * per wave; 0: per wave.  Same products in the same order:
                                  * bit-identical results.  Default 1. */
#define BEER_OPT_COUNT 7
int beer_hip_set_option(int option, int value);
int beer_hip_get_option(int option);   /* the value, or BEER_EINVAL for an unknown option */

/* How float32 models multiply on the matrix cores is chosen PER CALL (float64
 * models always use the exact fp64 MFMA): the `dtype` argument of
 * beer_mixtureset_estep is BEER_F32 for the default arithmetic, "bf16x3" -- every
 * fp32 operand is held EXACTLY as three bf16 pieces (3 x 8 significand bits = the
 * 24 of fp32, fp32's exponent range: no scaling, no range restrictions) and every
 * product is the six leading partial products on v_mfma_f32_16x16x32_bf16, fp32
 * accumulation: operands and accumulation are fp32's own, what a product drops is
 * <= 2^-23 of it (2^-25 typically, no systematic sign), at 2.7x the rate of the
 * fp32 pipe -- or BEER_F32 | BEER_EXACT for v_mfma_f32_16x16x4_f32, bitwise an
 * fmaf chain.  The packed hand-over calls below (beer_mixture_estep_packed, ...)
 * are bf16x3 by construction.  The library keeps no mode. */
#define BEER_EXACT 0x10

/* ------------------------------------------------------------------------
 * Exponential-family parameter kernels (once per VB iteration, K = number of
 * distributions in the set, D = feature dimension).
 * ---------------------------------------------------------------------- */

/* E_q[T(theta)], the "natural form" every E-step consumes.
 * Replaces NormalWishart.expected_sufficient_statistics
 * (beer/dists/normalwishart.py:170-210) reached through
 * ConjugateBayesianParameter.natural_form (beer/models/parameters.py:131-132).
 * mean [K,D], scale [K], scale_matrix [K,D,D], dof [K] -> out [K, D*D+D+2]. */
int beer_nw_expected_stats(int dtype, int K, int D, const void* mean,
                           const void* scale, const void* scale_matrix,
                           const void* dof, void* out, void* stream);
/* The two calls around this comment in one launch (they share the
 * factorisation of the scale matrix, and a VB iteration needs both: E[T] for
 * the E-step, the log-normaliser for the KL term): out [K, D*D+D+2],
 * log_norm [K]. */
int beer_nw_expected_stats_log_norm(int dtype, int K, int D, const void* mean,
                                    const void* scale, const void* scale_matrix,
                                    const void* dof, void* out, void* log_norm,
                                    void* stream);
/* NormalWishart.log_norm (normalwishart.py:219-236) -> out [K]. */
int beer_nw_log_norm(int dtype, int K, int D, const void* mean,
                     const void* scale, const void* scale_matrix,
                     const void* dof, void* out, void* stream);
/* NormalWishart.natural_parameters (normalwishart.py:242-269) -> [K,Q]. */
int beer_nw_natural(int dtype, int K, int D, const void* mean,
                    const void* scale, const void* scale_matrix,
                    const void* dof, void* out, void* stream);
/* NormalWishartStdParams.from_natural_parameters (normalwishart.py:110-141). */
int beer_nw_from_natural(int dtype, int K, int D, const void* eta, void* mean,
                         void* scale, void* scale_matrix, void* dof,
                         void* stream);

/* The M-step of a Normal-Wishart posterior in one launch:
 * from_natural_parameters (normalwishart.py:110-141) and, from the same
 * elimination, what the next iteration asks of the new posterior: `exp_stats`
 * [K, D*D+D+2] (expected_sufficient_statistics, normalwishart.py:170-210),
 * `log_norm` [K] (normalwishart.py:219-236) and, when `moments` is not NULL, the
 * moments of its expected Gaussian, [K, D + D*D] = (mean, E[Lambda]^-1 = W^-1 / nu).
 * One factorisation where the three separate calls make two and an inverse. */
int beer_nw_update(int dtype, int K, int D, const void* eta, void* mean, void* scale,
                   void* scale_matrix, void* dof, void* exp_stats, void* log_norm,
                   void* moments, void* stream);

/* NormalGamma (diagonal covariance), beer/dists/normalgamma.py:118-146,
 * 151-157, 163-180, 77-94.  mean [K,D], scale [K], shape [K], rates [K,D]. */
int beer_ng_expected_stats(int dtype, int K, int D, const void* mean,
                           const void* scale, const void* shape,
                           const void* rates, void* out, void* stream);
int beer_ng_log_norm(int dtype, int K, int D, const void* mean,
                     const void* scale, const void* shape, const void* rates,
                     void* out, void* stream);
int beer_ng_natural(int dtype, int K, int D, const void* mean,
                    const void* scale, const void* shape, const void* rates,
                    void* out, void* stream);
int beer_ng_from_natural(int dtype, int K, int D, const void* eta, void* mean,
                         void* scale, void* shape, void* rates, void* stream);

/* IsotropicNormalGamma, beer/dists/isonormalgamma.py:119-157, 163-168,
 * 174-192, 78-95.  mean [K,D], scale [K], shape [K], rate [K]. */
int beer_ing_expected_stats(int dtype, int K, int D, const void* mean,
                            const void* scale, const void* shape,
                            const void* rate, void* out, void* stream);
int beer_ing_log_norm(int dtype, int K, int D, const void* mean,
                      const void* scale, const void* shape, const void* rate,
                      void* out, void* stream);
int beer_ing_natural(int dtype, int K, int D, const void* mean,
                     const void* scale, const void* shape, const void* rate,
                     void* out, void* stream);
int beer_ing_from_natural(int dtype, int K, int D, const void* eta, void* mean,
                          void* scale, void* shape, void* rate, void* stream);

/* Dirichlet (set of S pdfs over G categories), beer/dists/dirichlet.py:
 * 106-128 (E[T]), 135-138 (log_norm), 144-159 (natural), 71-81 (from). */
int beer_dirichlet_expected_stats(int dtype, int S, int G, const void* conc,
                                  void* out, void* stream);
int beer_dirichlet_log_norm(int dtype, int S, int G, const void* conc,
                            void* out, void* stream);
int beer_dirichlet_natural(int dtype, int S, int G, const void* conc,
                           void* out, void* stream);
int beer_dirichlet_from_natural(int dtype, int S, int G, const void* eta,
                                void* conc, void* stream);
/* Truncated stick-breaking categorical (any truncation P; device, one workgroup: up to
 * 1024 sticks ranked in LDS, beyond that on the global arrays).
 * beer_sb_transform_stats: SBCategorical._transform_stats
 * (beer/models/categorical.py:106-118) -- the sticks are ordered by decreasing
 * count (stable), `ordering[r]` = category of stick r, and the counts [P] become
 * the Dirichlet statistics of the sticks [P,2] = (count_i, count_i + sum of the
 * counts ordered after i), stored in the categories' own order.
 * beer_sb_log_weights: SBCategorical._log_prob + the re-ordering of
 * expected_log_likelihood (categorical.py:120-131, 157-159) -- E[ln pi_i] =
 * E[ln v_i] + sum over the sticks before i of E[ln(1 - v)], [P] in the categories'
 * order; `log_1_v_sum` (nullable) = sum_i E[ln(1 - v_i)], the statistic of the
 * Gamma hyper-prior (categorical.py:203-209). */
int beer_sb_transform_stats(int dtype, int P, const void* counts, int64_t* ordering,
                            void* stats, void* stream);
int beer_sb_log_weights(int dtype, int P, const void* conc, const int64_t* ordering,
                        void* log_w, void* log_1_v_sum, void* stream);
/* E[ln pi] of S categoricals: the `eye -> sufficient_statistics -> stats @
 * E[T]` sequence of Mixture._log_weights (beer/models/mixture.py:45-48) and
 * MixtureSet._log_weights (mixtureset.py:64-67) -> out [S,G]. */
int beer_dirichlet_log_weights(int dtype, int S, int G, const void* conc,
                               void* out, void* stream);

/* Gamma (n independent pdfs), beer/dists/gamma.py:112-124, 129-131,
 * 137-140, 68-77.  shape [n], rate [n]; E[T] / natural are [2n]. */
int beer_gamma_expected_stats(int dtype, int n, const void* shape,
                              const void* rate, void* out, void* stream);
int beer_gamma_log_norm(int dtype, int n, const void* shape, const void* rate,
                        void* out /*[1]*/, void* stream);
int beer_gamma_natural(int dtype, int n, const void* shape, const void* rate,
                       void* out, void* stream);
int beer_gamma_from_natural(int dtype, int n, const void* eta, void* shape,
                            void* rate, void* stream);

/* KL(q || p) of K same-family members from their expected statistics,
 * natural parameters and log-normalisers: beer/dists/basedist.py:243-263.
 * out [K]. */
int beer_kl_div(int dtype, int K, int Q, const void* exp_stats_q,
                const void* eta_q, const void* eta_p, const void* lnorm_q,
                const void* lnorm_p, void* out, void* stream);

/* eta <- eta_q + lrate * (eta_p + stats - eta_q): the body of
 * ConjugateBayesianParameter.natural_grad_update
 * (beer/models/parameters.py:134-141).  n elements. */
int beer_natural_grad_step(int dtype, int64_t n, const void* eta_prior,
                           const void* eta_post, const void* stats,
                           double lrate, void* out, void* stream);

/* Dense statistics phi(X) [T,Q] exactly as the reference materialises them
 * (normalwishart.py:30-38 etc.).  Only for callers that ask for the tensor
 * (Model.sufficient_statistics(...).dense()); the E-step never forms it. */
int beer_suffstats_expand(int dtype, int cov, int64_t T, int D, const void* X,
                          void* out, void* stream);

/* ------------------------------------------------------------------------
 * E-step: expected log-likelihoods and responsibilities
 * ---------------------------------------------------------------------- */

/* Per-component expected log-likelihood + per-state mixture normaliser +
 * component responsibilities for S states of G Gaussians each (K = S*G):
 *   l[t,s,g]  = stat_scale * phi(x_t) . E[T]_{s,g} - D/2 ln 2pi
 *   w         = l + log_weights[s,g]
 *   log_norm[t,s]     = logsumexp_g w            (nullable)
 *   comp_resps[t,s,g] = exp(w - log_norm[t,s])   (nullable)
 *   pc_llh[t,s,g]     = l                         (nullable)
 * Replaces NormalLikelihood.__call__ (normalwishart.py:88-92; diag
 * normalgamma.py:55-59; iso isonormalgamma.py:56-60) through
 * NormalSet.expected_log_likelihood (beer/models/normalset.py:117-119),
 * Mixture.expected_log_likelihood (mixture.py:70-93: S = 1) and
 * MixtureSet.expected_log_likelihood (mixtureset.py:85-98).  With
 * `labels` (int64 [T], S must be 1) responsibilities are one-hot and
 * log_norm[t] = l[t, label] (mixture.py:85-87).  `log_weights` nullable
 * (= 0).  `stat_scale` reproduces HMM.posteriors' scaling of the statistics
 * (beer/models/hmm.py:119); pass 1.  `llh_sum` (nullable) += sum_t,s log_norm
 * in fp64.  With G > 1 the generic kernels need `comp_resps` as their work
 * buffer; the matrix-core path (workspace given, shape supported) keeps the
 * responsibilities in registers and accepts log_norm / llh_sum alone. */
int beer_mixtureset_estep(int dtype, int cov, int64_t T, int D, int S, int G,
                          const void* X, const void* exp_stats,
                          const void* log_weights, const int64_t* labels,
                          double stat_scale, void* pc_llh, void* log_norm,
                          void* comp_resps, double* llh_sum, void* workspace,
                          size_t workspace_bytes, void* stream);

/* Scratch the matrix-core (MFMA) implementations of the two calls around
 * this comment need (packed parameter / partial-sum images).  0 = the shape
 * has no MFMA implementation.  With `workspace` NULL or too small the calls
 * run the generic kernels -- same results, slower.  The workspace holds no
 * state between calls.  For float32 the size covers both arithmetic variants
 * (BEER_EXACT). */
size_t beer_estep_workspace_bytes(int dtype, int cov, int D, int S, int G);
size_t beer_accumulate_workspace_bytes(int dtype, int cov, int D, int S, int G);
/* ... the same for a call over T frames: never smaller; for float32 diagonal / isotropic
 * Gaussians accumulated without state posteriors (T >= 16384, D <= 64: acc_diag.hip) it also
 * holds the partial sums of every 2048-frame chain, which a second small kernel adds up in
 * fp64 -- with the smaller workspace that call adds them with fp64 atomics (same sums). */
size_t beer_accumulate_frames_workspace_bytes(int dtype, int cov, int64_t T, int D, int S, int G);

/* gamma-weighted sufficient statistics (N_k, sum r x, sum r xx^T) packed as
 * the reference packs them, [K,Q] = resps^T @ phi(X), accumulated (+=) in
 * fp64:  acc[k,:] += sum_t comp_resps[t,k] * state_resps[t, k / G] * phi(x_t).
 * Replaces NormalSet.accumulate (beer/models/normalset.py:121-123) with the
 * joint responsibilities of MixtureSet.accumulate (mixtureset.py:100-112);
 * `state_resps` [T,S] nullable (= 1), `comp_resps` [T,K] nullable (= 1).
 * float32 without BEER_EXACT, diagonal / isotropic, comp_resps given and no state_resps,
 * T >= 16384, D <= 64, 16 <= K <= 65536: the bf16x3 arithmetic of the E-step (acc_diag.hip:
 * three bf16 pieces per operand, six MFMAs per product, 512-frame sums on the matrix cores,
 * float32 to 2048 frames, fp64 beyond); everything else float32: the exact fp32 kernels. */
int beer_normal_accumulate(int dtype, int cov, int64_t T, int D, int S, int G,
                           const void* X, const void* comp_resps,
                           const void* state_resps, double* acc, void* workspace,
                           size_t workspace_bytes, void* stream);

/* The E-step -> accumulate hand-over of a single mixture (S = 1, float32, bf16x3
 * arithmetic) without the float32 responsibilities in between.  `packed_resps`
 * (beer_packed_resps_bytes(T, D, K) bytes: 6 bytes per element with T rounded up
 * to 64 and K to 128, then, for 129..256 components, the transposed frames the
 * E-step kernel leaves behind for the accumulation) receives each responsibility
 * already split into the three bf16 pieces the accumulation kernel multiplies
 * with (r = p0 + p1 + p2 exactly), laid out as that kernel's LDS tiles: per
 * (64 frames, 128 components) 48 KB = three planes of 16 KB, plane q = the pieces
 * p_q as rows [component][64 frames] whose 16-byte chunks sit at position
 * chunk ^ (component & 7); frames >= T and components >= K are 0.  The
 * accumulation kernel copies the tiles to LDS as they are (LDS-DMA).  Same
 * reference functions as the two calls above (mixture.py:86-101 for the
 * responsibilities, normalset.py:121-123 for the statistics).  beer_unpack_resps
 * restores the [T,K] float32 matrix exactly.  The accumulation workspace also
 * holds the frames transposed into 64-frame tiles, hence its own size query with T.
 * EINVAL: shape without a matrix-core path, workspace NULL / too small (sizes
 * from beer_estep_workspace_bytes(BEER_F32, ...) and
 * beer_accumulate_packed_workspace_bytes). */
size_t beer_packed_resps_bytes(int64_t T, int D, int K);
size_t beer_accumulate_packed_workspace_bytes(int cov, int64_t T, int D, int K);
int beer_mixture_estep_packed(int cov, int64_t T, int D, int K, const float* X,
                              const float* exp_stats, const float* log_weights,
                              float* log_norm, void* packed_resps, double* llh_sum,
                              void* workspace, size_t workspace_bytes, void* stream);
int beer_normal_accumulate_packed(int cov, int64_t T, int D, int K, const float* X,
                                  const void* packed_resps, double* acc,
                                  void* workspace, size_t workspace_bytes,
                                  void* stream);
/* The same hand-over for a mixture SET around the forward-backward pass (float32,
 * bf16x3 arithmetic, full covariance; S states of G components, G a power of two
 * in 8..128): beer_mixtureset_estep_packed leaves `log_norm` [T, S] and
 * the responsibilities WITHIN each state's mixture as packed tiles (layout above,
 * K = S * G; no frame tiles behind them); after the forward-backward pass
 * beer_mixtureset_accumulate_packed multiplies the state posteriors
 * `state_resps` [T, S] in while a tile sits in LDS -- (p0 + p1 + p2) * gamma in
 * fp32, split again, each element once per workgroup -- and accumulates
 *     acc[k,:] += sum_t r[t,k] state_resps[t, k / G] phi(x_t)           (fp64, +=)
 * MixtureSet.expected_log_likelihood / accumulate (beer/models/mixtureset.py:85-112)
 * around HMM.expected_log_likelihood (beer/models/hmm.py:73-92): neither the
 * [T, K] float32 responsibilities nor their product with the state posteriors is
 * ever stored.  beer_mixtureset_packed_supported: 1 where both calls have a
 * kernel; EINVAL elsewhere (callers then use beer_mixtureset_estep +
 * beer_normal_accumulate).  Workspaces: beer_estep_workspace_bytes(BEER_F32, ...)
 * and beer_mixtureset_accumulate_packed_workspace_bytes (grows with T: the
 * transposed frames and state posteriors). */
int beer_mixtureset_packed_supported(int cov, int D, int S, int G);
size_t beer_mixtureset_accumulate_packed_workspace_bytes(int cov, int64_t T, int D, int S,
                                                         int G);
int beer_mixtureset_estep_packed(int cov, int64_t T, int D, int S, int G, const float* X,
                                 const float* exp_stats, const float* log_weights,
                                 float* log_norm, void* packed_resps, double* llh_sum,
                                 void* workspace, size_t workspace_bytes, void* stream);
int beer_mixtureset_accumulate_packed(int cov, int64_t T, int D, int S, int G,
                                      const float* X, const void* packed_resps,
                                      const float* state_resps, double* acc,
                                      void* workspace, size_t workspace_bytes,
                                      void* stream);
/* The packed buffer from float32 responsibilities: comp_resps [T, S*G], times
 * state_resps[t, k / G] when given (the joint responsibilities of
 * MixtureSet.accumulate, mixtureset.py:100-112), split and tiled as above;
 * (S*G) % 4 == 0.  This is how float32 responsibilities that exist in memory
 * (labels, generic E-step) reach the bf16x3 accumulation kernel. */
int beer_pack_resps(int64_t T, int D, int S, int G, const float* X,
                    const float* comp_resps, const float* state_resps,
                    void* packed_resps, void* stream);
int beer_unpack_resps(int64_t T, int K, const void* packed_resps, float* resps,
                      void* stream);

/* The accumulation of a mixture set WITHOUT its responsibilities in memory
 * (float32, bf16x3 arithmetic; diagonal / isotropic covariances, D <= 64: few
 * statistics per Gaussian).  beer_mixtureset_estep is then called with
 * comp_resps = NULL and leaves only the per-state log-normalisers `log_norm`
 * [T, S]; after the forward-backward pass this call recomputes the component
 * logits on the matrix cores, forms
 *     r[t,k] * state_resps[t, k / G] = exp(l[t,k] - log_norm[t, k / G]) * state_resps[t, k / G]
 * in registers and multiplies them with the statistics of the frames at once:
 *     acc[k,:] += sum_t r[t,k] state_resps[t, k / G] phi(x_t)            (fp64, +=)
 * -- MixtureSet.accumulate (beer/models/mixtureset.py:100-112) over
 * NormalSet.accumulate (normalset.py:121-123) with the responsibilities of
 * mixtureset.py:85-98 recomputed instead of cached.  `exp_stats`, `log_weights`
 * ([S,G], nullable) as given to beer_mixtureset_estep; `state_resps` [T,S]
 * nullable (= 1).  At K = 1920 Gaussians this removes 15.4 GB of traffic per
 * million frames (the [T, K] matrix written and read back).
 * EINVAL: full covariance, D > 64, workspace NULL / too small. */
size_t beer_accumulate_fused_workspace_bytes(int cov, int D, int S, int G);
int beer_mixtureset_accumulate_fused(int cov, int64_t T, int D, int S, int G,
                                     const float* X, const float* exp_stats,
                                     const float* log_weights, const float* log_norm,
                                     const float* state_resps, const void* frame_image,
                                     double* acc, void* workspace, size_t workspace_bytes,
                                     void* stream);

/* `frame_image` above (nullable): the operands of that kernel that depend on the
 * frames only -- phi(x_t) = [x^2, x, 1] of every 32-frame tile as bf16x3 pieces, once in
 * the layout of the logits' A fragments and once in that of the statistics' B
 * fragments, 1056 bytes per frame at D = 40 (the statistics' data columns: the counts are
 * summed on the vector ALU).  The kernel is bound by vector
 * instruction issue and two thirds of its vector instructions rebuild these for each
 * of the K / 64 component chunks; the frames do not change between the VB iterations
 * of a training run, so a caller that keeps X resident builds the image once per block
 * of frames (beer_frame_image; the same T, D, cov as the accumulation call it is handed
 * to) and the accumulation loads every fragment with 16-byte loads.  Same results bit
 * for bit.  beer_frame_image_bytes = 0: no image for this shape (full covariance, D not a
 * multiple of 4 or > 40); the accumulation ignores an image it cannot use. */
size_t beer_frame_image_bytes(int cov, int64_t T, int D);
/* beer_mixtureset_estep(BEER_F32, ..., comp_resps = NULL) -- the per-state
 * log-normalisers of a mixture set (mixtureset.py:85-98), all that a forward-backward
 * pass needs -- with the A fragments of the logits taken from the same frame image:
 * no staging of the frames, no fragment arithmetic.  Same log_norm bit for bit.
 * Workspace as beer_estep_workspace_bytes(BEER_F32, cov, D, S, G).  EINVAL where the
 * image does not exist or G < 4. */
int beer_mixtureset_lognorm_image(int cov, int64_t T, int D, int S, int G, const float* X,
                                  const float* exp_stats, const float* log_weights,
                                  const void* frame_image, float* log_norm, double* llh_sum,
                                  void* workspace, size_t workspace_bytes, void* stream);
int beer_frame_image(int cov, int64_t T, int D, const float* X, void* image,
                     size_t image_bytes, void* stream);

/* Mixture-weight statistics from the accumulated Gaussian statistics: the
 * zero-order count is N_k = -2 * acc[k, Q-2]; out[s,g] = N_{s,g} for
 * g < G-1 and out[s,G-1] = sum_g N_{s,g} -- the "last column <- row sum"
 * convention of CategoricalLikelihood.sufficient_statistics
 * (beer/dists/dirichlet.py:18-21) summed over frames as in
 * Categorical.accumulate (categorical.py:78-79) and
 * CategoricalSet.accumulate_from_jointresps (categoricalset.py:57-58). */
int beer_weights_from_acc(int S, int G, int Q, const double* acc,
                          double* out /*[S,G] +=*/, void* stream);

/* ------------------------------------------------------------------------
 * HMM inference over a ragged batch of utterances
 * ---------------------------------------------------------------------- */

/* Inference graph in device memory (CompiledGraph, beer/graph.py:243-268):
 * dense init/final log-probabilities plus the finite transitions in two
 * CSR forms (by destination for the forward / Viterbi recursions, by source
 * for the backward recursion).  Arcs inside a row are sorted by the other
 * end's index (ascending) -- Viterbi's first-index tie-break relies on it.
 * Built on the host by beer_amd.graph.CompiledGraph; -inf transitions are
 * simply absent.  Rows are additionally cut into segments of at most
 * BEER_SEG arcs so that the log-sum-exp of a high-degree state (a phone
 * start in a phone loop has one incoming arc per phone) is reduced by
 * several lanes instead of one. */
#define BEER_SEG 8
typedef struct {
    int32_t n_states;
    int32_t n_arcs;
    int32_t n_in_seg;          /* segments (<= BEER_SEG arcs of one row) by destination */
    int32_t n_out_seg;         /* ... by source */
    const void* init;          /* [S]  */
    const void* final;         /* [S]  */
    const int32_t* in_ptr;     /* [S+1] arcs grouped by destination */
    const int32_t* in_src;     /* [nnz] source state               */
    const int32_t* in_dst;     /* [nnz] destination state (row id)  */
    const void* in_w;          /* [nnz] log-probability             */
    const int32_t* in_seg;     /* [n_in_seg+1] arc offsets of the segments  */
    const int32_t* in_row_seg; /* [S+1] segment range of every destination  */
    const int32_t* out_ptr;    /* [S+1] arcs grouped by source      */
    const int32_t* out_dst;    /* [nnz] destination state           */
    const int32_t* out_src;    /* [nnz] source state (row id)       */
    const void* out_w;         /* [nnz] */
    const int32_t* out_seg;    /* [n_out_seg+1] */
    const int32_t* out_row_seg;/* [S+1] */
    const struct beer_graph_lowdeg* lowdeg;   /* optional factorised image, see below */
} beer_graph;

/* Optional low-degree image of the same graph, used by the fast
 * forward-backward variant.  compile() of the reference eliminates
 * non-emitting states, which turns a phone loop's pivot state into a dense
 * P x P block of arcs (every phone end -> every phone start) whose
 * log-weights are rank one: A[e, s] = src_w[e] + dst_w[s].  Such a block is
 * kept here as a "hub" (one log-sum-exp over its sources per frame) and removed
 * from the CSR, which leaves every state with a handful of arcs.  Present only
 * when every remaining row has <= BEER_SEG arcs.  */
typedef struct beer_graph_lowdeg {
    int32_t n_arcs;            /* arcs left after removing the hub blocks */
    int32_t n_hubs;
    const int32_t* in_ptr;     /* [S+1] */
    const int32_t* in_src;     /* [n_arcs] */
    const void* in_w;          /* [n_arcs] */
    const int32_t* out_ptr;    /* [S+1] */
    const int32_t* out_dst;    /* [n_arcs] */
    const void* out_w;         /* [n_arcs] */
    const int32_t* hub_src_id; /* [S] hub fed by the state, -1: none */
    const void* hub_src_w;     /* [S] */
    const int32_t* hub_dst_id; /* [S] hub feeding the state, -1: none */
    const void* hub_dst_w;     /* [S] */
    const int32_t* src_ptr;    /* [n_hubs+1] source states of every hub ... */
    const int32_t* src_list;
    const int32_t* dst_ptr;    /* [n_hubs+1] ... and its destination states */
    const int32_t* dst_list;
} beer_graph_lowdeg;

/* Ragged batch: utterance u owns frames [frame_off[u], frame_off[u+1]) of the
 * packed feature matrix and rows [llh_off[u], ...) (in elements) of the
 * packed per-state buffers, whose row length is graphs[graph_id[u]].n_states.
 * All arrays are device arrays of length nutt (+1 for frame_off). */
typedef struct {
    int32_t nutt;
    int32_t max_states;        /* max n_states over the batch's graphs (LDS sizing) */
    int32_t max_arcs;          /* max n_arcs over the batch's graphs   (LDS sizing) */
    int32_t max_segs;          /* max(n_in_seg, n_out_seg) over the graphs          */
    int32_t all_lowdeg;        /* != 0: every graph has a `lowdeg` image            */
    int32_t n_graphs;
    const int64_t* frame_off;  /* [nutt+1] */
    const int64_t* llh_off;    /* [nutt]   element offsets */
    const int32_t* graph_id;   /* [nutt]   */
    const beer_graph* graphs;  /* device array of graph descriptors */
    const int32_t* pdf_off;    /* [n_graphs+1] offsets into pdf_ids */
    const int32_t* pdf_ids;    /* concatenated pdf_id_mapping of every graph */
    /* what the host knows about the `lowdeg` images (0 = unknown: the
     * one-wave-per-utterance recursion is then not used) */
    int32_t max_degree;        /* largest in / out degree of their CSRs (>= 1) */
    int32_t max_hubs;          /* largest n_hubs */
    int32_t max_hub_members;   /* largest number of sources / destinations of a hub */
    int32_t reserved;
    /* (nullable) [nutt]: the order in which the one-wave-per-utterance kernels hand the
     * utterances to their waves -- longest first, so that the four waves of a workgroup
     * finish together and the launch ends with its shortest utterances (the reference
     * loops over utterances in file order, accumulate.py:39-59; the sums do not care).
     * NULL: 0, 1, 2, ... */
    const int32_t* order;
} beer_batch;

/* pc_llhs[u][t,s] = scale * pc_all[frame_off[u]+t, pdf_id[s]]: the gather of
 * DynamicallyOrderedModelSet.expected_log_likelihood
 * (beer/models/modelset.py:140-146) and the acoustic scale of
 * HMM.expected_log_likelihood (beer/models/hmm.py:79). */
int beer_hmm_gather(int dtype, const beer_batch* batch_h, int S_total,
                    const void* pc_all, double scale, void* pc_llhs,
                    void* stream);

/* Log-space forward-backward, per-frame-normalised state posteriors and
 * (optional) summed transition posteriors:
 * CompiledGraph._baum_welch_forward/_backward/posteriors
 * (beer/graph.py:270-326).  gamma has the layout of pc_llhs.  `alpha_ws` is
 * caller-provided fp64 scratch with as many elements.  `xi_sum` (nullable, [S,S]
 * fp64, +=) receives sum_u sum_t xi_t -- every utterance must then use
 * graph 0; `gamma0_sum` (nullable, [S] fp64, +=) receives sum_u gamma_0.
 * `lognorm_mean` (nullable, [nutt]) receives mean_t lognorm_t
 * (graph.py:326).  When every graph of the batch carries a `lowdeg` image
 * the factorised recursion runs; transition posteriors through a hub are
 * then reported per destination state, summed over the hub's sources, in
 * `hub_flow` ([S] fp64, +=; required whenever xi_sum is given) instead of as
 * individual xi_sum entries.  `hub_ws` (nullable; fp64 scratch, 4 per frame of
 * the batch) lets low-degree graphs of at most 256 states run one WAVE per
 * utterance (no workgroup barrier in the recursion: about 4x faster). */
int beer_hmm_forward_backward(int dtype, const beer_batch* batch_h,
                              const void* pc_llhs, double* alpha_ws, double* hub_ws,
                              void* gamma, double* xi_sum, double* gamma0_sum,
                              double* hub_flow, void* lognorm_mean, void* stream);
/* How many utterances of the batch the last beer_hmm_forward_backward /
 * beer_hmm_posteriors_fused call on `hub_ws` ran in LOG SPACE (no reference counterpart:
 * beer/graph.py:270-326 is log-space throughout; the one-wave kernels run on scaled
 * probabilities and hand an utterance over to their log-space twin when a column or a
 * frame's normaliser falls below 2^-800 of its scale, or a log-likelihood is NaN).
 * `count` (device, int64) += that number.  EINVAL unless the batch is of the kind the
 * one-wave kernels take (other kernels are log-space: nothing to count). */
int beer_hmm_fb_log_count(const beer_batch* batch_h, const double* hub_ws, int64_t* count,
                          void* stream);

/* Doubles `hub_ws` of beer_hmm_forward_backward must hold BESIDES the hub values
 * (BEER_MAX_HUBS per frame) for this batch: 0 while the arc lists of its largest
 * graph fit a CU's LDS (about 4000 arcs with transition posteriors in float32);
 * beyond that the general kernel keeps its per-arc scratch there -- 2 max_arcs +
 * max_segs doubles for each of at most 512 workgroups -- and reads the topology
 * from the graph image, so that graphs of any density run (the reference's dense
 * recursion, graph.py:270-326, has no size limit); only the per-state arrays
 * stay in LDS (up to ~4000 states). */
size_t beer_hmm_fb_scratch_doubles(int dtype, const beer_batch* batch_h, int want_xi);

/* The HMM inference step of a whole shard in ONE launch, for batches whose
 * graphs all carry a `lowdeg` image with at most one hub of at most 64 sources /
 * destinations and have at most 256 states (phone loops with their pivot declared
 * as a hub, alignment chains): the gather of
 * beer_hmm_gather, the recursions of beer_hmm_forward_backward (one wave per
 * utterance) and the scatter of beer_hmm_scatter without the packed pc_llhs /
 * gamma arrays in between.
 *   pc_all       [n_frames, S_total] per-pdf log-likelihoods (read through the
 *                pdf ids, times `scale`: modelset.py:140-146, hmm.py:79)
 *   state_resps  [n_frames, S_total]: scale * gamma at the pdf ids
 *                (modelset.py:148-154, hmm.py:95).  atomic_out = 0: plain stores --
 *                every graph's pdf ids are distinct; the caller zero-fills the
 *                array unless they also cover 0 .. S_total-1.  atomic_out = 1:
 *                added atomically into the zero-filled array (repeated ids).
 *                atomic_out = 2 (S_total <= 512, else BEER_EINVAL): whole rows -- a
 *                wave adds its states' posteriors into a zero row in LDS (repeated
 *                ids add up there) and stores all S_total entries of the frame,
 *                zeros included: the array needs no initialisation and sees no
 *                atomic.  What alignment-graph training uses (accumulate.py:39-59
 *                with --alis: a transcription names a phone more than once).
 *   utt_llh      (nullable, [nutt] fp64, +=) sum_t sum_s gamma * scale * pc
 *                (hmm.py:87);  gamma0_sum, hub_flow: as above (graph 0 for all).
 *   frame_llh    (nullable, [n_frames] of dtype, stored) the per-frame value
 *                sum_s gamma_ts * scale * pc_ts that HMM.expected_log_likelihood returns
 *                (hmm.py:87), reduced over the wave while both factors are in registers --
 *                what beer_rowdot(state_resps, pc_all) computes from the two [T, S] arrays.
 *   alpha_ws     fp64 scratch, sum_u T_u * S_u;  hub_ws: fp64 scratch, n_frames.
 * EINVAL when the batch is not of that kind (use the three calls instead). */
int beer_hmm_posteriors_fused(int dtype, const beer_batch* batch_h, int S_total,
                              const void* pc_all, double scale, double* alpha_ws,
                              double* hub_ws, void* state_resps, int atomic_out,
                              double* gamma0_sum, double* hub_flow, double* utt_llh,
                              void* frame_llh, void* stream);

/* Per-frame transition posteriors in the reference's own layout, xi [T-1, S, S]
 * (beer/graph.py:308-323: normalised per frame, NaN -> 0), for ONE utterance,
 * from what beer_hmm_forward_backward left behind: `alpha` (its fp64 workspace
 * [T,S]), `gamma` [T,S], the emission log-likelihoods `llhs` [T,S] it was given
 * and the dense transition log-probabilities `trans` [S,S].  The batched path
 * never forms this tensor (8 S^2 bytes per frame); it exists for callers that
 * switch the reference layout on (beer_amd.reference_layout). */
int beer_hmm_trans_posteriors(int dtype, int64_t T, int S, const double* alpha,
                              const void* llhs, const void* gamma, const void* trans,
                              void* xi, void* stream);

/* Viterbi + backtrack, CompiledGraph.best_path (beer/graph.py:329-344):
 * first-index tie-break, -inf safe, int64 state path per frame (packed like
 * the frames).  `bt_ws` is int32 scratch with the layout of pc_llhs.  With `map_pdf` != 0 the path is mapped through pdf_id_mapping as
 * HMM.decode does (beer/models/hmm.py:105-114). */
int beer_hmm_viterbi(int dtype, const beer_batch* batch_h, const void* pc_llhs,
                     int32_t* bt_ws, int64_t* path, int map_pdf, void* stream);

/* One-hot posteriors of a given state path (hmm.py:42-58: viterbi=True /
 * state_path= branches): gamma one-hot, xi_sum[p_t, p_t+1] += 1,
 * gamma0_sum[p_0] += 1. */
int beer_hmm_path_posteriors(int dtype, const beer_batch* batch_h,
                             const int64_t* path, void* gamma, double* xi_sum,
                             double* gamma0_sum, void* stream);

/* Scatter the (scaled) state posteriors back to pdf ids, repeated ids add
 * (DynamicallyOrderedModelSet.accumulate, beer/models/modelset.py:148-154,
 * with the scale of HMM.accumulate, hmm.py:95), and the per-frame expected
 * log-likelihood exp_llh[t] = sum_s gamma[t,s] * pc_llhs[t,s] (hmm.py:87).
 * state_resps [Ttot, S_total] must be zeroed by the caller; `exp_llh`
 * nullable [Ttot]; `utt_llh` nullable fp64 [nutt] (+= sum_t exp_llh). */
int beer_hmm_scatter(int dtype, const beer_batch* batch_h, int S_total,
                     const void* pc_llhs, const void* gamma, double scale,
                     void* state_resps, void* exp_llh, double* utt_llh,
                     void* stream);

/* Per-utterance sums of a per-frame quantity: out[u] += sum_t v[t]
 * (the `exp_llh.sum()` of evidence_lower_bound,
 * beer/inference/objectives.py:184, for every utterance of a batch). */
int beer_segment_sum(int dtype, int32_t nutt, const int64_t* frame_off,
                     const void* v, double* out, void* stream);

/* ---- "statistics-in" E-step (VAE models, beer/models/vae.py:63-89) ---------
 * The prior of a VAE receives dense sufficient statistics [T, Q] (averages of
 * phi(z) over the samples of the latent variable) instead of frames, and its
 * expected log-likelihood is differentiated w.r.t. them. */

/* float32 products with Q >= 512 and T >= 8192 (full-covariance latent) run on
 * the matrix cores in the E-step's bf16x3 arithmetic (float32 operands held
 * exactly as three bf16 pieces, float32 accumulation; sums over frames in chains
 * of 4096 frames, fp64 between workgroups); everything else, and fp64, on an
 * LDS-tiled kernel that accumulates in fp64.  No library underneath.  These entry
 * points take BEER_F32 / BEER_F64 only: BEER_EXACT (the fp32 MFMA of the frame
 * kernels) has no counterpart here -- large float32 shapes are bf16x3, as
 * accurate as a float32 product, in either mode of the host layer. */

/* out[t,k] = sum_q stats[t,q] * exp_stats[k,q] + base   -> [T, K]
 * (ConjugateLikelihood.__call__, beer/dists/normalgamma.py:55-59; base is the
 * log base measure -D/2 ln 2pi). */
int beer_dense_llh(int dtype, int64_t T, int Q, int K, const void* stats,
                   const void* exp_stats, double base, void* out, void* stream);

/* Gradient of sum_t grad[t] * sum_k weights[t,k] * llh[t,k] w.r.t. stats:
 * out[t,q] = grad[t] * sum_k weights[t,k] * exp_stats[k,q]   -> [T, Q].
 * This is what autograd gives the reference for `(pc_llh * resps).sum(-1)`
 * with detached responsibilities (mixture.py:79,92; hmm.py:81-87).
 * `grad` nullable (= 1). */
int beer_dense_llh_backward(int dtype, int64_t T, int K, int Q,
                            const void* weights, const void* grad,
                            const void* exp_stats, void* out, void* stream);

/* acc[k,q] += sum_t weights[t,k] * state_resps[t, k / G] * stats[t,q], fp64
 * [K, Q] (NormalSet.accumulate, normalset.py:121-123, with the joint
 * responsibilities of MixtureSet.accumulate, mixtureset.py:103-106).
 * `state_resps` nullable [T, K / G]. */
int beer_dense_accumulate(int dtype, int64_t T, int K, int Q, int G,
                          const void* weights, const void* state_resps,
                          const void* stats, double* acc, void* stream);

/* out[t] = sum_k a[t,k] * b[t,k]  (exp_llh = (pc_llhs * resps).sum(-1)). */
int beer_rowdot(int dtype, int64_t T, int K, const void* a, const void* b,
                void* out, void* stream);

/* Per (frame, mixture) softmax over G components of pc_llh[T, S*G] +
 * log_weights[S, G] (nullable): log_norm [T, S] and resps [T, S*G], both
 * nullable (mixture.py:78-82, mixtureset.py:92-95). */
int beer_softmax_groups(int dtype, int64_t T, int S, int G, const void* pc_llh,
                        const void* log_weights, void* log_norm, void* resps,
                        void* stream);

/* out[t,:] = (1/ns) sum_s phi(X[t*ns + s]) -> [T, Q]: the sample-averaged
 * statistics a VAE hands to its prior (vae.py:73-74: sufficient_statistics of
 * the [T*ns, D] samples, reshape, mean over the samples) without the
 * [T*ns, Q] intermediate.  ns = 1 is beer_suffstats_expand. */
int beer_suffstats_mean(int dtype, int cov, int64_t T, int ns, int D,
                        const void* X, void* out, void* stream);

/* Backward of beer_suffstats_mean: grad_X[t*ns + s, :] =
 * (1/ns) J_phi(x_ts)^T grad_stats[t, :] -> [T*ns, D]
 * (normalwishart.py:30-38 / normalgamma.py:22-31 / isotropicnormalgamma.py
 * sufficient_statistics differentiated by autograd in the reference). */
int beer_suffstats_backward(int dtype, int cov, int64_t T, int ns, int D,
                            const void* X, const void* grad_stats,
                            void* grad_X, void* stream);

/* ---- one sample per frame: the prior of a VAE on the frame kernels -----------
 * With ONE sample z_t per frame (vae.py:63-86, nsamples = 1) the statistics the
 * prior receives are phi(z_t): its E-step and accumulation are the frame entry
 * points above (beer_mixtureset_estep, beer_normal_accumulate ...) on the samples,
 * and the gradient the reference's autograd carries from `(pc_llhs * resps).sum(-1)`
 * through the statistics to the samples (mixture.py:79,92; hmm.py:81-87;
 * normalwishart.py:30-38) is one product, with no [T, Q] tensor in between:
 *
 *   out[t,:] = grad[t] * sum_k weights[t,k] * d(phi(x_t) . exp_stats[k]) / d x_t  -> [T, D]
 *
 * = beer_suffstats_backward(beer_dense_llh_backward(weights, grad, exp_stats)) at
 * ns = 1.  `grad` nullable (= 1); `weights` [T, K]; `exp_stats` [K, Q] as for
 * beer_dense_llh.  BEER_F32 / BEER_F64.  Full covariance, float32, 8 <= D <= 64,
 * T >= 4096 with a workspace of beer_frames_llh_backward_workspace_bytes (the
 * parameters as bf16x3 MFMA fragments): matrix cores, bf16x3 arithmetic, float32
 * accumulation over the components; diagonal / isotropic, float32, D <= 128: one
 * [T, K] x [K, 2 D] product in the same arithmetic, combined with the frames in
 * its epilogue (HBM-bound).  Other shapes with T >= 4096 and that workspace
 * (<= 256 MiB): the two calls above over chunks of frames whose [frames, Q]
 * gradient fits it.  T < 4096 (the query returns 0), or workspace
 * NULL: one thread per output, float64 accumulation. */
size_t beer_frames_llh_backward_workspace_bytes(int dtype, int cov, int64_t T, int D, int K);
int beer_frames_llh_backward(int dtype, int cov, int64_t T, int D, int K, const void* X,
                             const void* weights, const void* grad, const void* exp_stats,
                             void* out, void* workspace, size_t workspace_bytes,
                             void* stream);

/* ---- feature front-end (beer/features.py, `beer features extract`) ----------
 * Ragged batch of utterances: `signal` holds the samples of all utterances
 * back to back, utterance u owns samples [sample_off[u], sample_off[u+1]) and
 * output rows [frame_off[u], frame_off[u+1]), with
 * frames_u = (samples_u - flen) / fstep + 1 (features.py:131,178).  Offsets
 * are DEVICE arrays of nutt + 1 int64.  All arithmetic is float64 as in the
 * reference (numpy), except the float32 pre-emphasis of `fbank()`. */

typedef struct {
    int32_t flen;        /* frame length in samples, int(srate * flen) */
    int32_t fstep;       /* frame shift in samples, int(srate * frate) */
    int32_t fft_len;     /* 2^(floor(log2(flen)) + 1), 64 .. 2048 */
    int32_t mode;        /* 0: features.fbank() -- float32 pre-emphasis over the
                            whole signal (features.py:182-184);
                            1: features.short_term_mspec() -- DC removal and
                            pre-emphasis inside each frame (features.py:124-139) */
    int32_t nfilters;    /* rows of `filters`; 0 = keep the fft_len/2 magnitudes */
    int32_t apply_log;   /* log(x + log_offset) after the filter bank */
    int32_t n_dct;       /* columns of `dct`; 0 = no cosine transform */
    int32_t add_energy;  /* prepend sum_f log-mel * norm (extract.py:148-151) */
    double preemph;
    double log_offset;   /* 1 in fbank() (features.py:204), 1e-6 in the CLI */
    double norm;         /* sqrt(2 / nfilters) (extract.py:131) */
    const double* window;   /* device [flen] */
    const double* filters;  /* device [nfilters, fft_len/2] (create_fbank) */
    const int32_t* filt_lo; /* device [nfilters] first non-zero bin, nullable */
    const int32_t* filt_hi; /* device [nfilters] last non-zero bin, nullable */
    const double* dct;      /* device [nfilters, n_dct] (extract.py:36-40) */
    const double* lifter;   /* device [n_dct], nullable (extract.py:141-145) */
} beer_feaconf;

/* mean[u] = mean of the samples of utterance u (the DC offset removed by
 * short_term_mspec, features.py:124).  in_dtype: BEER_I16 / BEER_F32 / BEER_F64. */
int beer_features_signal_mean(int in_dtype, int32_t nutt, const int64_t* sample_off,
                              const void* signal, double* mean, void* stream);

/* Framing -> pre-emphasis -> window -> |FFT| -> filter bank -> log ->
 * cosine transform * norm * lifter, with the optional energy column first.
 * out [total_frames, out_ld] float64, the features occupy the first
 * add_energy + (n_dct ? n_dct : nfilters ? nfilters : fft_len/2) columns of a
 * row (room for the deltas after them).  `utt_mean` nullable (mode 1 only). */
int beer_features_extract(int in_dtype, int32_t nutt, const int64_t* sample_off,
                          const int64_t* frame_off, int64_t total_frames,
                          const void* signal, const double* utt_mean,
                          const beer_feaconf* conf, double* out, int32_t out_ld,
                          void* stream);

/* One order of derivatives (features.add_deltas, features.py:95-105):
 * out[t,d] = sum_{j=-wlen..wlen} j / (2 sum j^2) * in[clamp(t+j), d], t clamped
 * inside its utterance.  `in` / `out` point at the first of D columns of rows
 * with stride ld (they may be column blocks of the same buffer). */
int beer_features_deltas(int32_t nutt, const int64_t* frame_off, int64_t total_frames,
                         int32_t D, int32_t ld, int32_t wlen, const double* in,
                         double* out, void* stream);

/* Per-utterance mean normalisation in place (extract.py:160-162). */
int beer_features_cmn(int32_t nutt, const int64_t* frame_off, int32_t D, int32_t ld,
                      double* x, void* stream);

/* Copy `nbytes` (a multiple of 16, both pointers 16-byte aligned) from PINNED
 * host memory to device memory with a kernel on `stream`.  Used for the small
 * batch / graph descriptors: a copy-engine transfer queued behind running
 * kernels was measured to stall the stream for tens of ms on this platform; a
 * kernel that reads the pinned pages directly does not. */
int beer_copy_pinned(void* dst_device, const void* src_pinned_host, size_t nbytes,
                     void* stream);

/* Measurement aid (bench.py's `clock`): one wave sleeps through `sleeps` x s_sleep 127 and
 * writes the elapsed ticks of the shader clock (s_memtime) and of the fixed 100 MHz reference
 * clock (s_memrealtime) to `ticks_out[0..1]` (device).  Launched on a side stream while the
 * kernels of an iteration run, their ratio x 100 MHz is the clock those kernels really ran at
 * -- what a roofline fraction measured on one box needs to be compared with another box's.
 * Replaces nothing of the reference (beer has no device clock to read). */
int beer_clock_probe(int64_t* ticks_out, int32_t sleeps, void* stream);

/* ---- graph compilation (HOST functions: host pointers, no stream) -----------
 * Graph.compile (beer/graph.py:185-240) and create_graph_from_seq
 * (beer/cli/subcommands/hmm/mkaligraph.py:18-39) in O(states + arcs), for one
 * graph or for a whole corpus of transcriptions at once.  The result lives in
 * an opaque host object; beer_graphset_image lays every graph out as the
 * beer_graph CSR image above inside ONE blob, for a single host-to-device
 * copy. */
typedef struct beer_graphset beer_graphset;

/* Compile one topology.  States 0..n_states-1 in insertion order, pdf_ids[s]
 * < 0 for a non-emitting state; arcs in insertion order, weights are
 * probabilities (already normalised by the caller, Graph.normalize). */
int beer_graph_compile(int32_t n_states, const int32_t* pdf_ids, int64_t n_arcs,
                       const int32_t* arc_src, const int32_t* arc_dst,
                       const double* arc_w, int32_t start_state,
                       int32_t end_state, beer_graphset** out);

/* Alignment graphs of n_utts transcriptions: utterance u is the sequence of
 * unit ids seq_units[seq_off[u] .. seq_off[u+1]), each unit a small HMM given
 * by its states (unit_pdf_ids, < 0: non-emitting), start / end state and arcs
 * (local state ids), all concatenated with *_off offsets.  Equals
 * create_graph_from_seq(seq, units) for every utterance: chain of unit
 * copies, Graph.normalize, Graph.compile. */
int beer_aligraphs_compile(int32_t n_units, const int32_t* unit_state_off,
                           const int32_t* unit_pdf_ids, const int32_t* unit_start,
                           const int32_t* unit_end, const int32_t* unit_arc_off,
                           const int32_t* unit_arc_src, const int32_t* unit_arc_dst,
                           const double* unit_arc_w, int64_t n_utts,
                           const int64_t* seq_off, const int32_t* seq_units,
                           beer_graphset** out);

int beer_graphset_free(beer_graphset* set);

/* Number of graphs and (nullable) cumulative state / arc offsets [n+1]. */
int beer_graphset_sizes(const beer_graphset* set, int64_t* n_graphs,
                        int64_t* state_off, int64_t* arc_off);

/* Concatenated contents (every pointer nullable): initial / final
 * probabilities and pdf ids per state, arcs sorted by (source, destination)
 * with local state ids and float32 probabilities, as the reference's tables
 * before `.log()`. */
int beer_graphset_export(const beer_graphset* set, float* init, float* fin,
                         int32_t* pdf_ids, int32_t* arc_src, int32_t* arc_dst,
                         float* arc_prob);

/* Size of, and the blob holding, the device image of every graph in `dtype`
 * (log-probabilities: float32 log of the float32 tables, as the reference).
 * `graphs[i]` receives the beer_graph descriptor of graph i whose pointers
 * are device_base + offset, i.e. valid once the blob has been copied to
 * device_base.  Graphs whose states all have <= BEER_SEG arcs get the
 * `lowdeg` image (no hubs). */
int beer_graphset_image_bytes(const beer_graphset* set, int dtype, size_t* bytes);
int beer_graphset_image(const beer_graphset* set, int dtype, void* host_blob,
                        uint64_t device_base, beer_graph* graphs);

#ifdef __cplusplus
}
#endif
#endif /* BEER_HIP_H */
