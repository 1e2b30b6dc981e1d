// Generic (any D, K, covariance type, fp32 / fp64) E-step kernels:
//   * per-component expected log-likelihood straight from the frames -- the
//     [T, Q] statistics tensor of the reference is never formed,
//   * per-state mixture normaliser + component responsibilities,
//   * responsibility-weighted sufficient statistics, accumulated in fp64.
// The MFMA fast path for full covariance lives in estep_mfma.hip and is
// selected by the C entry points below when its shape constraints hold.
//
// Reference restated: beer/models/normalset.py:117-123, mixture.py:70-102,
// mixtureset.py:85-112, dists/normalwishart.py:30-38,88-92 (and the
// normalgamma / isonormalgamma equivalents).

#include "common.h"
#include "estep_mfma.h"

using namespace beer;

namespace {

constexpr int kFrameTile = 64;      // frames per workgroup tile (lane = frame)
constexpr int kLlhThreads = 256;    // 4 waves, each walks a share of the comps
constexpr int kCompChunk = 64;      // components per workgroup

// ---------------------------------------------------------------------------
// Pass 1: w[t,k] = stat_scale * phi(x_t) . E[T]_k - D/2 ln2pi (+ log_weights)
// Lane = frame, the component's parameters are wave-uniform (scalar loads).
// Results are staged through LDS so that global writes are row-contiguous.
// ---------------------------------------------------------------------------
// With `fuse_G` > 0 (G divides the component chunk) the per-state logsumexp and
// the responsibilities are finished here, from the LDS tile, and the separate
// normalisation pass (one more read + write of the [T, K] buffer) is skipped.
template <typename T, int COV>
__global__ __launch_bounds__(kLlhThreads) void llh_kernel(
    int64_t nframes, int D, int K, const T* __restrict__ X, const T* __restrict__ expT,
    const T* __restrict__ logw, double stat_scale, T* __restrict__ pc_llh,
    T* __restrict__ w_out, int fuse_G, T* __restrict__ log_norm,
    double* __restrict__ llh_sum) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ldx = D + 1;
    T* xs = reinterpret_cast<T*>(smem);                 // [64][D+1]
    T* outs = xs + kFrameTile * ldx;                    // [64][kCompChunk+1]
    const int ldo = kCompChunk + 1;
    const int Q = stats_dim(COV, D);
    const int64_t t0 = (int64_t)blockIdx.x * kFrameTile;
    const int k0 = blockIdx.y * kCompChunk;
    const int nk = min(kCompChunk, K - k0);
    const int nt = (int)min<int64_t>(kFrameTile, nframes - t0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    for (int idx = tid; idx < nt * D; idx += kLlhThreads)
        xs[(idx / D) * ldx + idx % D] = X[t0 * D + idx];
    __syncthreads();

    const T* x = xs + lane * ldx;
    const double base = -0.5 * (double)D * kLog2Pi;
    if (lane < nt) {
        for (int kk = wave; kk < nk; kk += kLlhThreads / 64) {
            const T* e = expT + (size_t)(k0 + kk) * Q;
            double acc = 0.0;
            if (COV == BEER_FULL) {
                const T* L = e + D;
                for (int i = 0; i < D; ++i) {
                    T s = 0;
                    const T* Li = L + i * D;
                    for (int j = 0; j < D; ++j) s += Li[j] * x[j];
                    acc += (double)x[i] * ((double)e[i] - 0.5 * (double)s);
                }
            } else if (COV == BEER_DIAG) {
                for (int i = 0; i < D; ++i) {
                    const double xi = (double)x[i];
                    acc += xi * ((double)e[i] - 0.5 * (double)e[D + i] * xi);
                }
            } else {
                double x2 = 0.0;
                for (int i = 0; i < D; ++i) {
                    const double xi = (double)x[i];
                    acc += xi * (double)e[i];
                    x2 += xi * xi;
                }
                acc -= 0.5 * (double)e[D] * x2;
            }
            const double zero = (COV == BEER_ISO) ? 0.5 * (double)D : 0.5;
            acc += -0.5 * (double)e[Q - 2] + zero * (double)e[Q - 1];
            outs[lane * ldo + kk] = (T)(stat_scale * acc + base);
        }
    }
    __syncthreads();
    if (fuse_G == 0) {
        for (int idx = tid; idx < nt * nk; idx += kLlhThreads) {
            const int t = idx / nk, kk = idx % nk;
            const T v = outs[t * ldo + kk];
            const size_t o = (size_t)(t0 + t) * K + k0 + kk;
            if (pc_llh) pc_llh[o] = v;
            if (w_out) w_out[o] = logw ? (T)(v + logw[k0 + kk]) : v;
        }
        return;
    }
    __shared__ double red[8];
    const int G = fuse_G, ngrp = nk / G, S = K / G;
    double mine = 0.0;
    for (int idx = tid; idx < nt * ngrp; idx += kLlhThreads) {
        const int t = idx / ngrp, grp = idx % ngrp;
        const T* row = outs + t * ldo + grp * G;
        const T* lw = logw ? logw + k0 + grp * G : nullptr;
        T m = row[0] + (lw ? lw[0] : (T)0);
        for (int gi = 1; gi < G; ++gi) {
            const T w = row[gi] + (lw ? lw[gi] : (T)0);
            m = w > m ? w : m;
        }
        T ln = m;
        if (m != (T)-INFINITY && m != (T)INFINITY) {
            double sum = 0.0;
            for (int gi = 0; gi < G; ++gi)
                sum += exp((double)(row[gi] + (lw ? lw[gi] : (T)0)) - (double)m);
            ln = (T)((double)m + log(sum));
        }
        const size_t o = (size_t)(t0 + t) * K + k0 + grp * G;
        for (int gi = 0; gi < G; ++gi) {
            if (pc_llh) pc_llh[o + gi] = row[gi];
            if (w_out)
                w_out[o + gi] = (T)exp((double)(row[gi] + (lw ? lw[gi] : (T)0)) - (double)ln);
        }
        if (log_norm) log_norm[(size_t)(t0 + t) * S + (k0 / G) + grp] = ln;
        mine += (double)ln;
    }
    if (llh_sum) {
        const double tot = block_sum(mine, red);
        if (threadIdx.x == 0) atomicAdd(llh_sum, tot);
    }
}

// ---------------------------------------------------------------------------
// Pass 2: per (frame, state) logsumexp over the G components, responsibilities
// in place.  One thread per (t, s).
// ---------------------------------------------------------------------------
template <typename T>
__global__ void normalise_kernel(int64_t nframes, int S, int G, T* __restrict__ w,
                                 T* __restrict__ log_norm, bool write_resps,
                                 double* __restrict__ llh_sum) {
    __shared__ double red[8];
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double mine = 0.0;
    if (idx < nframes * S) {
        T* row = w + idx * G;
        T m = row[0];
        for (int g = 1; g < G; ++g) m = row[g] > m ? row[g] : m;
        T ln;
        if (m == -INFINITY || m == INFINITY) {
            ln = m;
        } else {
            double s = 0.0;
            for (int g = 0; g < G; ++g) s += exp((double)row[g] - (double)m);
            ln = (T)((double)m + log(s));
        }
        if (write_resps)
            for (int g = 0; g < G; ++g) row[g] = (T)exp((double)row[g] - (double)ln);
        if (log_norm) log_norm[idx] = ln;
        mine = (double)ln;
    }
    if (llh_sum) {
        const double tot = block_sum(mine, red);
        if (threadIdx.x == 0) atomicAdd(llh_sum, tot);
    }
}

// labels= branch of Mixture.expected_log_likelihood (mixture.py:85-87).
template <typename T>
__global__ void labels_kernel(int64_t nframes, int K, const int64_t* __restrict__ labels,
                              const T* __restrict__ pc, T* __restrict__ resps,
                              T* __restrict__ log_norm, double* __restrict__ llh_sum) {
    __shared__ double red[8];
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double mine = 0.0;
    if (t < nframes) {
        const int64_t lab = labels[t];
        const T v = pc[t * K + lab];          // read before the row is overwritten
        if (resps)
            for (int k = 0; k < K; ++k) resps[t * K + k] = (T)(k == lab ? 1 : 0);
        if (log_norm) log_norm[t] = v;
        mine = (double)v;
    }
    if (llh_sum) {
        const double tot = block_sum(mine, red);
        if (threadIdx.x == 0) atomicAdd(llh_sum, tot);
    }
}

// ---------------------------------------------------------------------------
// Accumulation: acc[k, q] += sum_t r[t,k] * phi_q(x_t).  Lane = statistic q,
// each lane keeps KB components in registers; the responsibilities are
// wave-uniform scalars.  fp32 partial sums are flushed to fp64 every frame
// tile, fp64 atomics at the end of the block's frame range.
// ---------------------------------------------------------------------------
constexpr int kAccThreads = 256;
constexpr int kAccKB = 8;
constexpr int kAccTile = 64;

template <typename T, int COV>
__global__ __launch_bounds__(kAccThreads) void accumulate_kernel(
    int64_t nframes, int D, int S, int G, const T* __restrict__ X,
    const T* __restrict__ comp_resps, const T* __restrict__ state_resps,
    int64_t frames_per_block, double* __restrict__ acc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* xs = reinterpret_cast<T*>(smem);                // [kAccTile][D]
    T* ws = xs + kAccTile * D;                          // [kAccTile][kAccKB]
    const int K = S * G, Q = stats_dim(COV, D);
    const int q = blockIdx.x * kAccThreads + threadIdx.x;
    const int k0 = blockIdx.y * kAccKB;
    const int nkb = min(kAccKB, K - k0);
    const int64_t tb = (int64_t)blockIdx.z * frames_per_block;
    const int64_t te = min(nframes, tb + frames_per_block);

    // which statistic this lane owns
    int qi = 0, qj = 0, kind;    // kind 0: x_i, 1: x_i*x_j*-.5, 2: -.5, 3: +.5 (iso .5 D), 4: -.5|x|^2
    if (q < D) { kind = 0; qi = q; }
    else if (q == Q - 2) kind = 2;
    else if (q == Q - 1) kind = 3;
    else if (q >= Q) kind = -1;
    else if (COV == BEER_FULL) { kind = 1; qi = (q - D) / D; qj = (q - D) % D; }
    else if (COV == BEER_DIAG) { kind = 1; qi = qj = q - D; }
    else kind = 4;

    double dacc[kAccKB];
#pragma unroll
    for (int b = 0; b < kAccKB; ++b) dacc[b] = 0.0;

    for (int64_t t0 = tb; t0 < te; t0 += kAccTile) {
        const int nt = (int)min<int64_t>(kAccTile, te - t0);
        __syncthreads();
        for (int idx = threadIdx.x; idx < nt * D; idx += kAccThreads) xs[idx] = X[t0 * D + idx];
        for (int idx = threadIdx.x; idx < nt * kAccKB; idx += kAccThreads) {
            const int t = idx / kAccKB, b = idx % kAccKB;
            T w = 0;
            if (b < nkb) {
                const int k = k0 + b;
                w = comp_resps ? comp_resps[(t0 + t) * K + k] : (T)1;
                if (state_resps) w *= state_resps[(t0 + t) * S + k / G];
            }
            ws[idx] = w;
        }
        __syncthreads();
        if (kind < 0) continue;
        T facc[kAccKB];
#pragma unroll
        for (int b = 0; b < kAccKB; ++b) facc[b] = 0;
        for (int t = 0; t < nt; ++t) {
            const T* x = xs + t * D;
            T phi;
            if (kind == 0) phi = x[qi];
            else if (kind == 1) phi = (T)-0.5 * (x[qi] * x[qj]);
            else if (kind == 2) phi = (T)-0.5;
            else if (kind == 3) phi = (COV == BEER_ISO) ? (T)(0.5 * D) : (T)0.5;
            else {
                T s = 0;
                for (int d = 0; d < D; ++d) s += x[d] * x[d];
                phi = (T)-0.5 * s;
            }
            const T* w = ws + t * kAccKB;
#pragma unroll
            for (int b = 0; b < kAccKB; ++b) facc[b] += w[b] * phi;
        }
#pragma unroll
        for (int b = 0; b < kAccKB; ++b) dacc[b] += (double)facc[b];
    }
    if (kind < 0) return;
#pragma unroll
    for (int b = 0; b < kAccKB; ++b)
        if (b < nkb) atomicAdd(acc + (size_t)(k0 + b) * Q + q, dacc[b]);
}

// One wave per state; lanes stride over its G components.
__global__ __launch_bounds__(64) void weights_from_acc_kernel(int S, int G, int Q,
                                                              const double* __restrict__ acc,
                                                              double* __restrict__ out) {
    const int s = blockIdx.x, lane = threadIdx.x;
    double tot = 0.0;
    for (int g = lane; g < G; g += 64) {
        const double n = -2.0 * acc[(size_t)(s * G + g) * Q + Q - 2];
        tot += n;
        if (g < G - 1) out[s * G + g] += n;
    }
    tot = wave_sum(tot);
    if (lane == 0) out[s * G + G - 1] += tot;
}

template <typename T>
__global__ void segment_sum_kernel(const int64_t* __restrict__ frame_off,
                                   const T* __restrict__ v, double* __restrict__ out) {
    __shared__ double red[8];
    const int u = blockIdx.x;
    const int64_t b = frame_off[u], e = frame_off[u + 1];
    double s = 0.0;
    for (int64_t t = b + threadIdx.x; t < e; t += blockDim.x) s += (double)v[t];
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[u] += s;
}

// ---- launchers -------------------------------------------------------------

template <typename T, int COV>
int llh_launch(int64_t nframes, int D, int K, const void* X, const void* expT, const void* logw,
               double stat_scale, void* pc_llh, void* w_out, int fuse_G, void* log_norm,
               double* llh_sum, hipStream_t s) {
    const size_t lds = ((size_t)kFrameTile * (D + 1) + (size_t)kFrameTile * (kCompChunk + 1)) * sizeof(T);
    const int64_t tiles = (nframes + kFrameTile - 1) / kFrameTile;
    // gridDim.x is 2^31-1 on gfx950; frames beyond that would need a loop.
    BEER_REQUIRE(tiles < (int64_t)2147483647);
    const dim3 grid((unsigned)tiles, (unsigned)((K + kCompChunk - 1) / kCompChunk));
    hipLaunchKernelGGL((llh_kernel<T, COV>), grid, dim3(kLlhThreads), lds, s, nframes, D, K,
                       (const T*)X, (const T*)expT, (const T*)logw, stat_scale, (T*)pc_llh,
                       (T*)w_out, fuse_G, (T*)log_norm, llh_sum);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

template <typename T>
int estep_launch(int cov, int64_t nframes, int D, int S, int G, const void* X, const void* expT,
                 const void* logw, const int64_t* labels, double stat_scale, void* pc_llh,
                 void* log_norm, void* comp_resps, double* llh_sum, void* ws, size_t ws_bytes,
                 void* stream, bool exact) {
    BEER_REQUIRE(nframes >= 0 && D >= 1 && S >= 1 && G >= 1 && cov >= 0 && cov <= 2);
    BEER_REQUIRE(X && expT);
    BEER_REQUIRE(!labels || S == 1);
    if (nframes == 0) return BEER_OK;
    hipStream_t s = as_stream(stream);
    const int K = S * G;
    // The normaliser works in place on the responsibilities buffer; without one
    // (pc_llh only, or G == 1 log_norm only) pass 1 alone is enough.
    const bool need_norm = (log_norm || comp_resps || llh_sum) && !labels;
    void* w_buf = comp_resps;
    // group-aligned shapes on every arithmetic; any G on the split path when only the
    // log-normalisers are wanted (groups padded to a power of two)
    // (the exact kernels take D <= 96 / 64, the bf16x3 kernels D <= 128)
    const bool aligned = beer_mfma::supported_llh(D, S, G, sizeof(T)) &&
                         ws_bytes >= beer_mfma::estep_workspace_bytes(sizeof(T), cov, D, S, G);
    const bool split_ws = sizeof(T) == 4 && !exact &&
                          ws_bytes >= beer_mfma::estepx_workspace_bytes(cov, D, S, G) &&
                          beer_mfma::estepx_workspace_bytes(cov, D, S, G) > 0;
    const bool aligned_x = split_ws && beer_mfma::supported_llh_x(D, S, G);
    const bool padded = split_ws && !comp_resps && beer_mfma::supported_llh_split(D, S, G);
    const bool use_x = aligned_x || padded;
    const bool mfma_ok = !labels && !pc_llh && stat_scale == 1.0 && ws && (aligned || use_x) &&
                         (log_norm || comp_resps || llh_sum);
    if (need_norm && !w_buf && !mfma_ok) {
        // G == 1: log_norm == w; let pass 1 write into log_norm directly (the
        // matrix-core path keeps the responsibilities in registers and needs no
        // buffer for any G).
        BEER_REQUIRE(G == 1 && log_norm);
        w_buf = log_norm;
    }
    int rc;
    void* pc_arg = labels ? (pc_llh ? pc_llh : comp_resps) : pc_llh;
    BEER_REQUIRE(!labels || pc_arg);
    void* w_arg = labels ? nullptr : (need_norm ? w_buf : nullptr);

    if (mfma_ok) {
        // gfx950 matrix-core path: GEMM + (grouped) softmax fused, one kernel
        if (use_x)
            return beer_mfma::estep_bf16x3(cov, nframes, D, S, G, (const float*)X,
                                           (const float*)expT, (const float*)logw,
                                           (float*)comp_resps, (float*)log_norm, llh_sum, ws,
                                           ws_bytes, s);
        return sizeof(T) == 4
                   ? beer_mfma::estep_f32(cov, nframes, D, S, G, (const float*)X,
                                          (const float*)expT, (const float*)logw,
                                          (float*)comp_resps, (float*)log_norm, llh_sum, ws,
                                          ws_bytes, s)
                   : beer_mfma::estep_f64(cov, nframes, D, S, G, (const double*)X,
                                          (const double*)expT, (const double*)logw,
                                          (double*)comp_resps, (double*)log_norm, llh_sum, ws,
                                          ws_bytes, s);
    }
    // generic kernels; the normalisation is fused when a component chunk holds
    // whole states
    const bool fuse = need_norm && !labels && (kCompChunk % G == 0) && (comp_resps || G == 1);
    const int fuse_G = fuse ? G : 0;
    void* ln_arg = fuse ? log_norm : nullptr;
    double* sum_arg = fuse ? llh_sum : nullptr;
    if (fuse && w_buf == log_norm) w_arg = nullptr;      // G == 1, no resps wanted
    if (cov == BEER_FULL)
        rc = llh_launch<T, BEER_FULL>(nframes, D, K, X, expT, logw, stat_scale, pc_arg, w_arg,
                                      fuse_G, ln_arg, sum_arg, s);
    else if (cov == BEER_DIAG)
        rc = llh_launch<T, BEER_DIAG>(nframes, D, K, X, expT, logw, stat_scale, pc_arg, w_arg,
                                      fuse_G, ln_arg, sum_arg, s);
    else
        rc = llh_launch<T, BEER_ISO>(nframes, D, K, X, expT, logw, stat_scale, pc_arg, w_arg,
                                     fuse_G, ln_arg, sum_arg, s);
    if (rc != BEER_OK) return rc;
    if (fuse) return BEER_OK;

    if (labels) {
        const T* pc = (const T*)pc_arg;
        // when pc aliases comp_resps the kernel reads pc[t,label] before it
        // overwrites the row: one thread owns the whole row.
        hipLaunchKernelGGL(labels_kernel<T>, dim3((unsigned)((nframes + 255) / 256)), dim3(256), 0,
                           s, nframes, K, labels, pc, (T*)comp_resps, (T*)log_norm, llh_sum);
        BEER_LAUNCH_CHECK();
        return BEER_OK;
    }
    if (need_norm) {
        const int64_t n = nframes * S;
        const bool alias = (w_buf == log_norm);
        hipLaunchKernelGGL(normalise_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                           nframes, S, G, (T*)w_buf, alias ? (T*)nullptr : (T*)log_norm,
                           comp_resps != nullptr, llh_sum);
        BEER_LAUNCH_CHECK();
    }
    return BEER_OK;
}

template <typename T, int COV>
int acc_launch(int64_t nframes, int D, int S, int G, const void* X, const void* cr,
               const void* sr, double* acc, hipStream_t s) {
    const int K = S * G, Q = stats_dim(COV, D);
    const int gx = (Q + kAccThreads - 1) / kAccThreads, gy = (K + kAccKB - 1) / kAccKB;
    // aim for ~4 workgroups per CU, at least 512 frames per workgroup -- one frame tile (64) for
    // inputs so small that the launch is a handful of workgroups either way: a lane walks its
    // workgroup's frames one by one, and the notebook-sized mixture (1000 frames, 8 x 6 statistics:
    // BASELINE config 1) spent 66 of its iteration's 142 us in two workgroups of 512 frames
    int64_t gz = (256LL * 4 + (int64_t)gx * gy - 1) / ((int64_t)gx * gy);
    const int64_t min_frames = nframes >= 65536 ? 512 : kAccTile;
    const int64_t max_z = (nframes + min_frames - 1) / min_frames;
    if (gz > max_z) gz = max_z;
    if (gz < 1) gz = 1;
    if (gz > 65535) gz = 65535;
    int64_t fpb = (nframes + gz - 1) / gz;
    fpb = (fpb + kAccTile - 1) / kAccTile * kAccTile;
    gz = (nframes + fpb - 1) / fpb;
    const size_t lds = ((size_t)kAccTile * D + (size_t)kAccTile * kAccKB) * sizeof(T);
    hipLaunchKernelGGL((accumulate_kernel<T, COV>), dim3(gx, gy, (unsigned)gz), dim3(kAccThreads),
                       lds, s, nframes, D, S, G, (const T*)X, (const T*)cr, (const T*)sr, fpb, acc);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

template <typename T>
int accumulate_launch(int cov, int64_t nframes, int D, int S, int G, const void* X,
                      const void* cr, const void* sr, double* acc, void* ws, size_t ws_bytes,
                      void* stream, bool exact) {
    BEER_REQUIRE(nframes >= 0 && D >= 1 && S >= 1 && G >= 1 && cov >= 0 && cov <= 2);
    BEER_REQUIRE(X && acc);
    if (nframes == 0) return BEER_OK;
    hipStream_t s = as_stream(stream);
    // (float32 responsibilities [T, K] in memory: the exact fp32 / fp64 MFMA kernels;
    // the bf16x3 accumulation takes packed tiles -- beer_pack_resps +
    // beer_normal_accumulate_packed -- whose size depends on T)
    // ... except diagonal / isotropic Gaussians without state posteriors (a set of single
    // Gaussians under an HMM: the prior of a VAE): few statistics per Gaussian, the float32
    // weights are split inside the kernel (acc_diag.hip)
    if (sizeof(T) == 4 && !exact && cr && !sr &&
        beer_mfma::supported_acc_diag(cov, nframes, D, S * G))
        return beer_mfma::acc_diag_bf16x3(cov, nframes, D, S * G, (const float*)X,
                                          (const float*)cr, acc, ws, ws_bytes, s);
    if (cr && ws && beer_mfma::supported_acc(D, S * G, sizeof(T)) &&
        ws_bytes >= beer_mfma::acc_workspace_bytes(cov, D, S * G, sizeof(T))) {
        return sizeof(T) == 4
                   ? beer_mfma::acc_f32(cov, nframes, D, S, G, (const float*)X, (const float*)cr,
                                        (const float*)sr, acc, ws, ws_bytes, s)
                   : beer_mfma::acc_f64(cov, nframes, D, S, G, (const double*)X,
                                        (const double*)cr, (const double*)sr, acc, ws, ws_bytes,
                                        s);
    }
    if (cov == BEER_FULL) return acc_launch<T, BEER_FULL>(nframes, D, S, G, X, cr, sr, acc, s);
    if (cov == BEER_DIAG) return acc_launch<T, BEER_DIAG>(nframes, D, S, G, X, cr, sr, acc, s);
    return acc_launch<T, BEER_ISO>(nframes, D, S, G, X, cr, sr, acc, s);
}

template <typename T>
int segment_sum_launch(int32_t nutt, const int64_t* frame_off, const void* v, double* out,
                       void* stream) {
    BEER_REQUIRE(nutt >= 0);
    if (nutt == 0) return BEER_OK;
    hipLaunchKernelGGL(segment_sum_kernel<T>, dim3(nutt), dim3(256), 0, as_stream(stream),
                       frame_off, (const T*)v, out);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

}  // namespace

extern "C" {

int beer_mixtureset_estep(int dtype, int cov, int64_t T, int D, int S, int G, const void* X,
                          const void* exp_stats, const void* log_weights,
                          const int64_t* labels, double stat_scale, void* pc_llh,
                          void* log_norm, void* comp_resps, double* llh_sum, void* workspace,
                          size_t workspace_bytes, void* stream) {
    const bool exact = (dtype & BEER_EXACT) != 0;
    BEER_DISPATCH(dtype & ~BEER_EXACT, estep_launch, cov, T, D, S, G, X, exp_stats, log_weights,
                  labels, stat_scale, pc_llh, log_norm, comp_resps, llh_sum, workspace,
                  workspace_bytes, stream, exact);
}

size_t beer_estep_workspace_bytes(int dtype, int cov, int D, int S, int G) {
    if (cov < 0 || cov > 2) return 0;
    dtype &= ~BEER_EXACT;
    const size_t exact = beer_mfma::estep_workspace_bytes(dtype == BEER_F64 ? 8 : 4, cov, D, S, G);
    if (dtype == BEER_F64) return exact;
    const size_t split = beer_mfma::estepx_workspace_bytes(cov, D, S, G);
    return exact > split ? exact : split;                 // either fp32 mode fits
}

size_t beer_accumulate_workspace_bytes(int dtype, int cov, int D, int S, int G) {
    if (cov < 0 || cov > 2) return 0;
    return beer_mfma::acc_workspace_bytes(cov, D, S * G, (dtype & ~BEER_EXACT) == BEER_F64 ? 8 : 4);
}

size_t beer_accumulate_frames_workspace_bytes(int dtype, int cov, int64_t T, int D, int S, int G) {
    const size_t base = beer_accumulate_workspace_bytes(dtype, cov, D, S, G);
    if (dtype != BEER_F32 || T < 0 || S < 1 || G < 1) return base;
    const size_t rows = beer_mfma::acc_diag_workspace_bytes(cov, T, D, S * G);
    return rows > base ? rows : base;
}

int beer_normal_accumulate(int dtype, int cov, int64_t T, int D, int S, int G, const void* X,
                           const void* comp_resps, const void* state_resps, double* acc,
                           void* workspace, size_t workspace_bytes, void* stream) {
    const bool exact = (dtype & BEER_EXACT) != 0;
    BEER_DISPATCH(dtype & ~BEER_EXACT, accumulate_launch, cov, T, D, S, G, X, comp_resps,
                  state_resps, acc, workspace, workspace_bytes, stream, exact);
}

int beer_mixture_estep_packed(int cov, int64_t T, int D, int K, const float* X,
                              const float* exp_stats, const float* log_weights, float* log_norm,
                              void* packed_resps, double* llh_sum, void* workspace,
                              size_t workspace_bytes, void* stream) {
    BEER_REQUIRE(T >= 0 && D >= 1 && K >= 1 && cov >= 0 && cov <= 2);
    BEER_REQUIRE(X && exp_stats && log_weights && packed_resps && workspace);
    BEER_REQUIRE(beer_mfma::supported_llh_x(D, 1, K));
    BEER_REQUIRE(workspace_bytes >= beer_mfma::estepx_workspace_bytes(cov, D, 1, K));
    if (T == 0) return BEER_OK;
    return beer_mfma::estep_bf16x3(cov, T, D, 1, K, X, exp_stats, log_weights,
                                   reinterpret_cast<float*>(packed_resps), log_norm, llh_sum,
                                   workspace, workspace_bytes, as_stream(stream), true);
}

int beer_normal_accumulate_packed(int cov, int64_t T, int D, int K, const float* X,
                                  const void* packed_resps, double* acc, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    BEER_REQUIRE(T >= 0 && D >= 1 && K >= 1 && cov >= 0 && cov <= 2);
    BEER_REQUIRE(X && packed_resps && acc && workspace);
    BEER_REQUIRE(beer_mfma::supported_acc_x(D, K));
    BEER_REQUIRE(workspace_bytes >= beer_mfma::accx_workspace_bytes(cov, T, D, K));
    if (T == 0) return BEER_OK;
    return beer_mfma::acc_bf16x3_packed(cov, T, D, K, X, packed_resps, acc, workspace,
                                        workspace_bytes, as_stream(stream));
}

int beer_mixtureset_packed_supported(int cov, int D, int S, int G) {
    return cov >= 0 && cov <= 2 && D >= 1 && D <= beer_mfma::kMaxDimX && S >= 1 && G >= 1 &&
           beer_mfma::supported_llh_packed_sets(cov, D, S, G) &&
           beer_mfma::supported_acc_sets(cov, D, S, G);
}

size_t beer_mixtureset_accumulate_packed_workspace_bytes(int cov, int64_t T, int D, int S, int G) {
    if (T < 0 || !beer_mixtureset_packed_supported(cov, D, S, G)) return 0;
    return beer_mfma::accxs_workspace_bytes(cov, T, D, S, G);
}

int beer_mixtureset_estep_packed(int cov, int64_t T, int D, int S, int G, const float* X,
                                 const float* exp_stats, const float* log_weights,
                                 float* log_norm, void* packed_resps, double* llh_sum,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    BEER_REQUIRE(T >= 0 && D >= 1 && S >= 1 && G >= 1 && cov >= 0 && cov <= 2);
    BEER_REQUIRE(X && exp_stats && packed_resps && workspace);
    BEER_REQUIRE(beer_mixtureset_packed_supported(cov, D, S, G));
    BEER_REQUIRE(workspace_bytes >= beer_mfma::estepx_workspace_bytes(cov, D, S, G));
    if (T == 0) return BEER_OK;
    return beer_mfma::estep_bf16x3(cov, T, D, S, G, X, exp_stats, log_weights,
                                   reinterpret_cast<float*>(packed_resps), log_norm, llh_sum,
                                   workspace, workspace_bytes, as_stream(stream), true);
}

int beer_mixtureset_accumulate_packed(int cov, int64_t T, int D, int S, int G, const float* X,
                                      const void* packed_resps, const float* state_resps,
                                      double* acc, void* workspace, size_t workspace_bytes,
                                      void* stream) {
    BEER_REQUIRE(T >= 0 && D >= 1 && S >= 1 && G >= 1 && cov >= 0 && cov <= 2);
    BEER_REQUIRE(X && packed_resps && state_resps && acc && workspace);
    BEER_REQUIRE(beer_mixtureset_packed_supported(cov, D, S, G));
    BEER_REQUIRE(workspace_bytes >= beer_mfma::accxs_workspace_bytes(cov, T, D, S, G));
    if (T == 0) return BEER_OK;
    return beer_mfma::acc_bf16x3_packed(cov, T, D, S * G, X, packed_resps, acc, workspace,
                                        workspace_bytes, as_stream(stream), S, G, state_resps);
}

size_t beer_accumulate_fused_workspace_bytes(int cov, int D, int S, int G) {
    if (cov < 0 || cov > 2 || D < 1 || S < 1 || G < 1) return 0;
    return beer_mfma::accf_workspace_bytes(cov, D, S, G);
}

int beer_mixtureset_lognorm_image(int cov, int64_t T, int D, int S, int G, const float* X,
                                  const float* exp_stats, const float* log_weights,
                                  const void* frame_image, float* log_norm, double* llh_sum,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    BEER_REQUIRE(T >= 0 && D >= 1 && S >= 1 && G >= 1 && cov >= 0 && cov <= 2);
    BEER_REQUIRE(frame_image && workspace && log_norm && (T == 0 || (X && exp_stats)));
    if (T == 0) return BEER_OK;
    return beer_mfma::estep_bf16x3(cov, T, D, S, G, X, exp_stats, log_weights, nullptr, log_norm,
                                   llh_sum, workspace, workspace_bytes, as_stream(stream), false,
                                   frame_image);
}

size_t beer_frame_image_bytes(int cov, int64_t T, int D) {
    if (cov < 0 || cov > 2 || T < 0 || D < 1) return 0;
    return beer_mfma::frame_image_bytes(cov, T, D);
}

int beer_frame_image(int cov, int64_t T, int D, const float* X, void* image, size_t image_bytes,
                     void* stream) {
    BEER_REQUIRE(cov >= 0 && cov <= 2 && T >= 0 && D >= 1);
    const size_t need = beer_mfma::frame_image_bytes(cov, T, D);
    BEER_REQUIRE(need > 0 && image && image_bytes >= need && (T == 0 || X));
    return beer_mfma::frame_image(cov, T, D, X, image, as_stream(stream));
}

int beer_mixtureset_accumulate_fused(int cov, int64_t T, int D, int S, int G, const float* X,
                                     const float* exp_stats, const float* log_weights,
                                     const float* log_norm, const float* state_resps,
                                     const void* frame_image, double* acc, void* workspace,
                                     size_t workspace_bytes, void* stream) {
    BEER_REQUIRE(T >= 0 && D >= 1 && S >= 1 && G >= 1 && cov >= 0 && cov <= 2);
    BEER_REQUIRE(beer_mfma::supported_accf(cov, D, S, G));
    BEER_REQUIRE(workspace && workspace_bytes >= beer_mfma::accf_workspace_bytes(cov, D, S, G));
    if (T == 0) return BEER_OK;
    BEER_REQUIRE(X && exp_stats && log_norm && acc);
    return beer_mfma::acc_fused_bf16x3(cov, T, D, S, G, X, exp_stats, log_weights, log_norm,
                                       state_resps, frame_image, acc, workspace, workspace_bytes,
                                       as_stream(stream));
}

size_t beer_packed_resps_bytes(int64_t T, int D, int K) {
    return T < 0 || D < 1 || D > beer_mfma::kMaxDimX || K < 1 ? 0 : beer_mfma::packed_resps_bytes(T, D, K);
}

size_t beer_accumulate_packed_workspace_bytes(int cov, int64_t T, int D, int K) {
    if (cov < 0 || cov > 2 || T < 0 || D < 1 || K < 1) return 0;
    return beer_mfma::accx_workspace_bytes(cov, T, D, K);
}

int beer_pack_resps(int64_t T, int D, int S, int G, const float* X, const float* comp_resps,
                    const float* state_resps, void* packed_resps, void* stream) {
    BEER_REQUIRE(T >= 0 && D >= 1 && D <= beer_mfma::kMaxDimX && S >= 1 && G >= 1 && (S * G) % 4 == 0);
    BEER_REQUIRE(T == 0 || (X && comp_resps && packed_resps));
    return beer_mfma::pack_resps(T, D, S, G, X, comp_resps, state_resps, packed_resps,
                                 as_stream(stream));
}

int beer_unpack_resps(int64_t T, int K, const void* packed_resps, float* resps, void* stream) {
    BEER_REQUIRE(T >= 0 && K >= 1 && (T == 0 || (packed_resps && resps)));
    return beer_mfma::unpack_resps(T, K, packed_resps, resps, as_stream(stream));
}

int beer_weights_from_acc(int S, int G, int Q, const double* acc, double* out, void* stream) {
    BEER_REQUIRE(S >= 0 && G >= 1 && Q >= 3 && acc && out);
    if (S == 0) return BEER_OK;
    hipLaunchKernelGGL(weights_from_acc_kernel, dim3(S), dim3(64), 0,
                       as_stream(stream), S, G, Q, acc, out);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

int beer_segment_sum(int dtype, int32_t nutt, const int64_t* frame_off, const void* v,
                     double* out, void* stream) {
    BEER_DISPATCH(dtype, segment_sum_launch, nutt, frame_off, v, out, stream);
}

}  // extern "C"
