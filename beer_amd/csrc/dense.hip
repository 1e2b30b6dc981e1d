// "Statistics-in" variants of the E-step: the caller hands dense sufficient
// statistics [T, Q] instead of frames.  This is the path of the VAE models
// (beer/models/vae.py:63-86): the prior model receives sample-averaged
// statistics of the latent variable, its expected log-likelihood must be
// differentiable w.r.t. those statistics (only sum_k r_k l_k carries gradient,
// mixture.py:79,92), and it accumulates resps^T @ stats.
//
// With dense statistics these are plain GEMMs with a small inner or outer
// dimension (Q = 2 D + 2 = 130 for a 64-d diagonal latent): HBM-bound on the
// [T, Q] and [T, K] operands.  One generic LDS-tiled kernel serves the three
// products; it is written for bandwidth (coalesced 64 x 16 tiles), not for MFMA.
//
// Reference restated: beer/dists/normalgamma.py:55-59 (llh = stats @ E[T]^T +
// log base measure), beer/models/normalset.py:121-123 (resps^T @ stats).

#include "common.h"
#include "estep_tiles.h"                 // split3: a float32 pair as three words of bf16 pieces

using namespace beer;

namespace {

constexpr int TM = 64, TN = 64, TK = 16;

// C[m,n] = alpha_m * sum_k A(m,k) B(k,n) + beta   (or atomically += into fp64)
// A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn].  grid.z splits k.
template <typename T, typename TC, bool ATOMIC>
__global__ __launch_bounds__(256) void gemm_kernel(int64_t M, int N, int64_t Kd, const T* A,
                                                   int64_t sam, int64_t sak, const T* B,
                                                   int64_t sbk, int64_t sbn, TC* C,
                                                   const T* row_scale, double beta,
                                                   int64_t k_per_block, const T* a_scale,
                                                   int a_group, int a_ld) {
    __shared__ T As[TK][TM + 1];
    __shared__ T Bs[TK][TN + 1];
    const int tid = threadIdx.x;
    const int64_t m0 = (int64_t)blockIdx.x * TM;
    const int n0 = blockIdx.y * TN;
    const int64_t kb = (int64_t)blockIdx.z * k_per_block;
    const int64_t ke = min(Kd, kb + k_per_block);
    const int tx = tid & 15, ty = tid >> 4;              // 16 x 16 threads, 4 x 4 outputs each
    double acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
    for (int64_t k0 = kb; k0 < ke; k0 += TK) {
        // the faster-varying global index goes on consecutive threads
        for (int idx = tid; idx < TM * TK; idx += 256) {
            int mm, kk;
            if (sak == 1) { kk = idx % TK; mm = idx / TK; } else { mm = idx % TM; kk = idx / TM; }
            const int64_t m = m0 + mm, k = k0 + kk;
            T a = (m < M && k < ke) ? A[m * sam + k * sak] : (T)0;
            // joint responsibilities: A(m, k) *= a_scale[k, m / a_group]
            if (a_scale && m < M && k < ke) a *= a_scale[k * a_ld + m / a_group];
            As[kk][mm] = a;
        }
        for (int idx = tid; idx < TN * TK; idx += 256) {
            int nn, kk;
            if (sbk == 1) { kk = idx % TK; nn = idx / TK; } else { nn = idx % TN; kk = idx / TN; }
            const int n = n0 + nn;
            const int64_t k = k0 + kk;
            Bs[kk][nn] = (n < N && k < ke) ? B[k * sbk + n * sbn] : (T)0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < TK; ++kk) {
            T a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = As[kk][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Bs[kk][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] += (double)a[i] * (double)b[j];
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t m = m0 + ty * 4 + i;
        if (m >= M) continue;
        const double sc = row_scale ? (double)row_scale[m] : 1.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n >= N) continue;
            if (ATOMIC) atomicAdd(reinterpret_cast<double*>(C) + m * N + n, acc[i][j]);
            else C[m * N + n] = (TC)(sc * acc[i][j] + beta);
        }
    }
}

template <typename T>
__global__ void rowdot_kernel(int64_t T_, int K, const T* __restrict__ a,
                              const T* __restrict__ b, T* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (t >= T_) return;
    const int lane = threadIdx.x & 63;
    double s = 0.0;
    for (int k = lane; k < K; k += 64) s += (double)a[t * K + k] * (double)b[t * K + k];
    s = wave_sum(s);
    if (lane == 0) out[t] = (T)s;
}

// w = pc + log_weights; per (t, s): log_norm = logsumexp_g w, resps = exp(w - log_norm)
template <typename T>
__global__ void softmax_groups_kernel(int64_t T_, int S, int G, const T* __restrict__ pc,
                                      const T* __restrict__ logw, T* __restrict__ log_norm,
                                      T* __restrict__ resps) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= T_ * S) return;
    const int s = (int)(idx % S);
    const T* row = pc + idx * G;
    const T* lw = logw ? logw + (size_t)s * G : nullptr;
    double m = -__builtin_huge_val();
    for (int g = 0; g < G; ++g) {
        const double w = (double)row[g] + (lw ? (double)lw[g] : 0.0);
        m = w > m ? w : m;
    }
    double ln = m;
    if (m > -__builtin_huge_val() && m < __builtin_huge_val()) {
        double sum = 0.0;
        for (int g = 0; g < G; ++g) sum += exp((double)row[g] + (lw ? (double)lw[g] : 0.0) - m);
        ln = m + log(sum);
    }
    if (log_norm) log_norm[idx] = (T)ln;
    if (resps)
        for (int g = 0; g < G; ++g)
            resps[idx * G + g] = (T)exp((double)row[g] + (lw ? (double)lw[g] : 0.0) - ln);
}

// out[t,:] = mean over the ns samples of frame t of phi(x_ts)
template <typename T>
__global__ void suffstats_mean_kernel(int cov, int64_t T_, int ns, int D,
                                      const T* __restrict__ X, T* __restrict__ out) {
    const int Q = stats_dim(cov, D);
    const int64_t total = T_ * Q;
    const double inv = 1.0 / ns;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = idx / Q;
        const int q = (int)(idx % Q);
        const T* x = X + t * ns * D;
        double v = 0.0;
        if (q >= Q - 2) {
            v = (q == Q - 2) ? -0.5 : (cov == BEER_ISO ? 0.5 * D : 0.5);
        } else if (q < D) {
            for (int s = 0; s < ns; ++s) v += (double)x[s * D + q];
            v *= inv;
        } else if (cov == BEER_FULL) {
            const int i = (q - D) / D, j = (q - D) % D;
            for (int s = 0; s < ns; ++s) v += (double)x[s * D + i] * (double)x[s * D + j];
            v *= -0.5 * inv;
        } else if (cov == BEER_DIAG) {
            const int i = q - D;
            for (int s = 0; s < ns; ++s) v += (double)x[s * D + i] * (double)x[s * D + i];
            v *= -0.5 * inv;
        } else {
            for (int s = 0; s < ns; ++s)
                for (int d = 0; d < D; ++d) v += (double)x[s * D + d] * (double)x[s * D + d];
            v *= -0.5 * inv;
        }
        __builtin_nontemporal_store((T)v, out + idx);
    }
}

// d phi(x) / dx contracted with the upstream gradient of the statistics; with
// ns > 1 the statistics of frame t are the mean over its ns samples (rows
// t*ns .. t*ns+ns-1 of X).
template <typename T>
__global__ void suffstats_backward_kernel(int cov, int64_t T_, int ns, int D,
                                          const T* __restrict__ X, const T* __restrict__ gs,
                                          T* __restrict__ gx) {
    const int Q = stats_dim(cov, D);
    const double inv = 1.0 / ns;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < T_ * ns * D;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = idx / D;
        const int d = (int)(idx % D);
        const T* x = X + n * D;
        const T* g = gs + (n / ns) * Q;
        double v = (double)g[d];
        if (cov == BEER_FULL) {
            double s = 0.0;
            for (int j = 0; j < D; ++j)
                s += ((double)g[D + d * D + j] + (double)g[D + j * D + d]) * (double)x[j];
            v -= 0.5 * s;
        } else if (cov == BEER_DIAG) {
            v -= (double)g[D + d] * (double)x[d];
        } else {
            v -= (double)g[D] * (double)x[d];
        }
        gx[idx] = (T)(v * inv);
    }
}

// ---------------------------------------------------------------------------
// Large float32 products (full-covariance latent: Q = D*D + D + 2 in the
// thousands): plain GEMMs with a long inner or outer dimension, where the LDS-tiled
// VALU kernel above runs at 25 TFLOP/s.  They run on the matrix cores in the same
// arithmetic as the E-step (estep_bf16.hip): every float32 operand held exactly as
// three bf16 pieces, the six leading partial products per element on
// v_mfma_f32_16x16x32_bf16, float32 accumulation -- float32 products to rounding
// level, no library underneath.
//
// One kernel serves the three products, C[m,n] = sum_k A(m,k) B(k,n) with arbitrary
// strides.  A workgroup (4 waves) owns a tile of 16 WM WAVES_M x 16 WN WAVES_N
// outputs and walks the inner dimension 32 at a time: every thread fetches quads of
// A and B (four consecutive k of one row; consecutive lanes along whichever index is
// contiguous in memory) one step ahead into registers, splits them ONCE into pieces
// and writes three bf16 planes to LDS (rows of 32 + 8 bf16: 16-byte chunks at an odd
// stride, conflict-free for the fragment reads); a wave reads its WM + WN fragments
// per plane and issues 6 WM WN MFMAs.
//
// Sums over frames (ATOMIC): the matrix core truncates when it aligns its addends
// (estep_bf16.hip, "chains"), so a chain of positive products drifts low by about
// 2^-25 per 32 frames.  A workgroup therefore sums at most kG3Chain frames (grid.z
// splits them) and adds its float32 sums to the fp64 accumulator with atomics.
// ---------------------------------------------------------------------------
constexpr int kBigMinQ = 512;
constexpr int64_t kBigMinT = 8192;
constexpr int kG3K = 32, kG3LD = 40;        // bf16 per LDS row (32 of them padding-free + 8)
constexpr int kG3Chain = 4096;

inline bool big_shape(int64_t T_, int Q, int K) {
    return Q >= kBigMinQ && T_ >= kBigMinT && K >= 1;
}

typedef unsigned int g3u4 __attribute__((ext_vector_type(4)));
typedef unsigned int g3u2 __attribute__((ext_vector_type(2)));
typedef __bf16 g3bf8 __attribute__((ext_vector_type(8)));

// AK / BK: A / B is contiguous along k (else along its row index m / n).  Rows past
// M and columns past N are fetched from the last valid one (their outputs are never
// stored); only the last, partial step of the inner dimension checks k.
template <int WM, int WN, int WAVES_M, int WAVES_N, bool AK, bool BK, bool ATOMIC>
__global__ __launch_bounds__(256, 2) void gemm3_kernel(
    int64_t M, int64_t N, int64_t Kd, const float* __restrict__ A, int sam, int sak,
    const float* __restrict__ B, int sbk, int sbn, void* __restrict__ Cout,
    const float* __restrict__ row_scale, float beta, int64_t k_per_block,
    const float* __restrict__ a_scale, int a_group, int a_ld) {
    static_assert(WAVES_M * WAVES_N == 4, "4 waves");
    constexpr int TMx = 16 * WM * WAVES_M, TNx = 16 * WN * WAVES_N;
    constexpr int QA = TMx * 8 / 256, QB = TNx * 8 / 256;          // quads per thread and step
    __shared__ __attribute__((aligned(16))) unsigned short As[3][TMx][kG3LD];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[3][TNx][kG3LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int64_t m0 = (int64_t)blockIdx.x * TMx, n0 = (int64_t)blockIdx.y * TNx;
    const int64_t kb = (int64_t)blockIdx.z * k_per_block;
    const int64_t ke = Kd < kb + k_per_block ? Kd : kb + k_per_block;
    const int ask = AK ? 1 : sak, bsk = BK ? 1 : sbk;

    // quad i of this thread: row of the tile, first k of the step, element offset of
    // (row, k = 0) from the tile's corner (32 bits: the host checks the strides)
    int arow[QA], ak[QA], brow[QB], bk[QB];
    unsigned aoff[QA], boff[QB], soff[QA];
#pragma unroll
    for (int i = 0; i < QA; ++i) {
        const int idx = tid + 256 * i;
        if (AK) { ak[i] = (idx & 7) * 4; arow[i] = idx >> 3; }
        else { arow[i] = idx % TMx; ak[i] = (idx / TMx) * 4; }
        const int64_t m = m0 + arow[i] < M ? m0 + arow[i] : M - 1;
        aoff[i] = (unsigned)(m - m0) * (unsigned)sam;
        soff[i] = a_scale ? (unsigned)(m / a_group) : 0u;
    }
#pragma unroll
    for (int i = 0; i < QB; ++i) {
        const int idx = tid + 256 * i;
        if (BK) { bk[i] = (idx & 7) * 4; brow[i] = idx >> 3; }
        else { brow[i] = idx % TNx; bk[i] = (idx / TNx) * 4; }
        const int64_t n = n0 + brow[i] < N ? n0 + brow[i] : N - 1;
        boff[i] = (unsigned)(n - n0) * (unsigned)sbn;
    }
    float pa[QA][4], pb[QB][4];
    auto fetch = [&](int64_t k0) {
        const float* __restrict__ Ab = A + m0 * sam + k0 * ask;
        const float* __restrict__ Bb = B + n0 * sbn + k0 * bsk;
        const float* __restrict__ Sb = a_scale ? a_scale + k0 * a_ld : nullptr;
        if (k0 + kG3K <= ke) {
#pragma unroll
            for (int i = 0; i < QA; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) pa[i][j] = Ab[aoff[i] + (unsigned)((ak[i] + j) * ask)];
            if (a_scale) {
#pragma unroll
                for (int i = 0; i < QA; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        pa[i][j] *= Sb[soff[i] + (unsigned)((ak[i] + j) * a_ld)];
            }
#pragma unroll
            for (int i = 0; i < QB; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) pb[i][j] = Bb[boff[i] + (unsigned)((bk[i] + j) * bsk)];
        } else {
            const int last = (int)(ke - 1 - k0);
#pragma unroll
            for (int i = 0; i < QA; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = ak[i] + j, kc = k < last ? k : last;
                    float v = Ab[aoff[i] + (unsigned)(kc * ask)];
                    if (a_scale) v *= Sb[soff[i] + (unsigned)(kc * a_ld)];
                    pa[i][j] = k <= last ? v : 0.f;
                }
#pragma unroll
            for (int i = 0; i < QB; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = bk[i] + j, kc = k < last ? k : last;
                    const float v = Bb[boff[i] + (unsigned)(kc * bsk)];
                    pb[i][j] = k <= last ? v : 0.f;
                }
        }
    };
    auto put = [&]() {
#pragma unroll
        for (int i = 0; i < QA; ++i) {
            unsigned lo[3], hi[3];
            beer_mfma::split3(pa[i][0], pa[i][1], lo);
            beer_mfma::split3(pa[i][2], pa[i][3], hi);
#pragma unroll
            for (int q = 0; q < 3; ++q)
                *reinterpret_cast<g3u2*>(&As[q][arow[i]][ak[i]]) = g3u2{lo[q], hi[q]};
        }
#pragma unroll
        for (int i = 0; i < QB; ++i) {
            unsigned lo[3], hi[3];
            beer_mfma::split3(pb[i][0], pb[i][1], lo);
            beer_mfma::split3(pb[i][2], pb[i][3], hi);
#pragma unroll
            for (int q = 0; q < 3; ++q)
                *reinterpret_cast<g3u2*>(&Bs[q][brow[i]][bk[i]]) = g3u2{lo[q], hi[q]};
        }
    };

    beer_mfma::f32x4 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = beer_mfma::f32x4{0.f, 0.f, 0.f, 0.f};
    const int fi = lane & 15, fg = lane >> 4;
    if (kb < ke) fetch(kb);
    for (int64_t k0 = kb; k0 < ke; k0 += kG3K) {
        __syncthreads();                       // the previous step's fragments are read
        put();
        __syncthreads();
        if (k0 + kG3K < ke) fetch(k0 + kG3K);
        // smallest products first; with (B fragment, A fragment) as the MFMA's (first,
        // second) operand a lane holds C[m = tile row fi][n = 4 fg + e]
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
        g3u4 af[WM][3], bf[WN][3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
            for (int i = 0; i < WM; ++i)
                af[i][q] = *reinterpret_cast<const g3u4*>(&As[q][(wm * WM + i) * 16 + fi][fg * 8]);
#pragma unroll
            for (int j = 0; j < WN; ++j)
                bf[j][q] = *reinterpret_cast<const g3u4*>(&Bs[q][(wn * WN + j) * 16 + fi][fg * 8]);
        }
#pragma unroll
        for (int pr = 0; pr < 6; ++pr)
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(g3bf8, bf[j][PB[pr]]),
                        __builtin_bit_cast(g3bf8, af[i][PA[pr]]), acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int64_t m = m0 + (wm * WM + i) * 16 + fi;
        if (m >= M) continue;
        const float sc = row_scale ? row_scale[m] : 1.f;
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int64_t nb = n0 + (wn * WN + j) * 16 + fg * 4;
            if (!ATOMIC && nb + 3 < N) {
                // the lane's 4 consecutive columns as ONE 16-byte store (rows of odd length are
                // only 4-byte aligned: the type says so); the 4 lanes that share a row fill a
                // 64-byte segment.  Dword stores, 4 per tile and 16 bytes apart within a row, made
                // the [T, Q] gradient of a full-covariance VAE prior leave at 0.85 TB/s.
                typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
                f4u v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = sc * acc[i][j][e] + beta;
                *reinterpret_cast<f4u*>(reinterpret_cast<float*>(Cout) + m * N + nb) = v;
                continue;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (nb + e >= N) continue;
                if (ATOMIC)
                    atomicAdd(reinterpret_cast<double*>(Cout) + m * N + nb + e,
                              (double)acc[i][j][e]);
                else
                    reinterpret_cast<float*>(Cout)[m * N + nb + e] = sc * acc[i][j][e] + beta;
            }
        }
    }
}

// Do the strides fit the kernel's 32-bit offsets inside a tile?
inline bool gemm3_fits(int64_t sam, int64_t sak, int64_t sbk, int64_t sbn, int64_t a_ld) {
    const int64_t lim = (int64_t)1 << 22;             // x 128 rows (or 32 k) x 4 bytes < 2^31
    return sam < lim && sak < lim && sbk < lim && sbn < lim && a_ld < lim;
}

// C [M, N] float32 = row_scale[m] * (A B) + beta, or (acc64) acc64 [M, N] += A B
template <bool AK, bool BK>
int gemm3(int64_t M, int64_t N, int64_t Kd, const float* A, int64_t sam, int64_t sak,
          const float* B, int64_t sbk, int64_t sbn, float* C, double* acc64,
          const float* row_scale, double beta, const float* a_scale, int a_group, int a_ld,
          hipStream_t s) {
#define BEER_G3(WM_, WN_, WVM, WVN)                                                            \
    do {                                                                                       \
        constexpr int TMx = 16 * WM_ * WVM, TNx = 16 * WN_ * WVN;                              \
        const int64_t gx = (M + TMx - 1) / TMx, gy = (N + TNx - 1) / TNx;                      \
        if (acc64) {                                                                           \
            /* frames over grid.z: chains of at most kG3Chain frames, and among those the  */ \
            /* split that leaves the fewest idle workgroup slots (2 per CU) in the last round */\
            const int64_t min_z = (Kd + kG3Chain - 1) / kG3Chain, slots = 512;                 \
            int64_t gz = min_z, best = INT64_MAX;                                              \
            for (int64_t z = min_z; z <= 2 * min_z || z * gx * gy <= 2 * slots; ++z) {         \
                if (z * kG3K > Kd && z > min_z) break;                                         \
                const int64_t cost = ((z * gx * gy + slots - 1) / slots) * ((Kd + z - 1) / z); \
                if (cost < best) { best = cost; gz = z; }                                      \
            }                                                                                  \
            int64_t kpb = (Kd + gz - 1) / gz;                                                  \
            kpb = (kpb + kG3K - 1) / kG3K * kG3K;                                              \
            gz = (Kd + kpb - 1) / kpb;                                                         \
            hipLaunchKernelGGL((gemm3_kernel<WM_, WN_, WVM, WVN, AK, BK, true>),               \
                               dim3((unsigned)gx, (unsigned)gy, (unsigned)gz), dim3(256), 0, s,\
                               M, N, Kd, A, (int)sam, (int)sak, B, (int)sbk, (int)sbn,         \
                               (void*)acc64, row_scale, 0.f, kpb, a_scale, a_group, a_ld);     \
        } else {                                                                               \
            hipLaunchKernelGGL((gemm3_kernel<WM_, WN_, WVM, WVN, AK, BK, false>),              \
                               dim3((unsigned)gx, (unsigned)gy, 1), dim3(256), 0, s, M, N, Kd, \
                               A, (int)sam, (int)sak, B, (int)sbk, (int)sbn, (void*)C,         \
                               row_scale, (float)beta, Kd, a_scale, a_group, a_ld);            \
        }                                                                                      \
    } while (0)
    if (M <= 64) BEER_G3(4, 2, 1, 4);           // 64 x 128
    else if (N <= 64) BEER_G3(2, 4, 4, 1);      // 128 x 64
    else BEER_G3(4, 4, 2, 2);                   // 128 x 128
#undef BEER_G3
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

// Full covariance: the upstream gradient of a frame holds a D x D block G, and
// every output needs row d AND column d of it -- read straight from memory by a
// thread per output (the kernel above) the rows are 4-byte gathers D * 4 bytes
// apart: 33 ms per 1 M frames at D = 64.  Here a wave owns a frame: it copies G
// (coalesced) into LDS rows of D + 1 words (conflict-free by row and by column),
// then lane d forms sum_j (G[d][j] + G[j][d]) x[j] for each of the frame's samples.
template <typename T>
__global__ __launch_bounds__(256) void suffstats_backward_full_kernel(
    int64_t T_, int ns, int D, const T* __restrict__ X, const T* __restrict__ gs,
    T* __restrict__ gx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int LD = D + 1, Q = D * D + D + 2;
    T* G = reinterpret_cast<T*>(smem) + wave * (D * LD + 2 * D);
    T* lin = G + D * LD;
    T* xs = lin + D;
    const double inv = 1.0 / ns;
    for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < T_; t += (int64_t)gridDim.x * 4) {
        const T* g = gs + t * Q;
        for (int e = lane; e < D * D; e += 64) {
            const int r = e / D, c = e - r * D;
            G[r * LD + c] = g[D + e];
        }
        if (lane < D) lin[lane] = g[lane];
        for (int sidx = 0; sidx < ns; ++sidx) {
            const int64_t n = t * ns + sidx;
            if (lane < D) xs[lane] = X[n * D + lane];
            __builtin_amdgcn_wave_barrier();
            if (lane < D) {
                double acc = 0.0;
                for (int j = 0; j < D; ++j)
                    acc += ((double)G[lane * LD + j] + (double)G[j * LD + lane]) * (double)xs[j];
                gx[n * D + lane] = (T)(((double)lin[lane] - 0.5 * acc) * inv);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

template <typename T>
int gemm_plain(int64_t M, int N, int64_t Kd, const T* A, int64_t sam, int64_t sak, const T* B,
               int64_t sbk, int64_t sbn, T* C, const T* row_scale, double beta, hipStream_t s) {
    const dim3 grid((unsigned)((M + TM - 1) / TM), (unsigned)((N + TN - 1) / TN), 1);
    hipLaunchKernelGGL((gemm_kernel<T, T, false>), grid, dim3(256), 0, s, M, N, Kd, A, sam, sak, B,
                       sbk, sbn, C, row_scale, beta, Kd, (const T*)nullptr, 1, 0);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

template <typename T>
int dense_llh_launch(int64_t T_, int Q, int K, const void* stats, const void* expT, double base,
                     void* out, void* stream) {
    BEER_REQUIRE(T_ >= 0 && Q >= 1 && K >= 1 && stats && expT && out);
    if (T_ == 0) return BEER_OK;
    // out[t,k] = sum_q stats[t,q] expT[k,q] + base
    if (sizeof(T) == 4 && big_shape(T_, Q, K) && gemm3_fits(Q, 1, 1, Q, 0))
        return gemm3<true, true>(T_, K, Q, (const float*)stats, Q, 1, (const float*)expT, 1, Q, (float*)out,
                     nullptr, nullptr, base, nullptr, 1, 0, as_stream(stream));
    return gemm_plain<T>(T_, K, Q, (const T*)stats, Q, 1, (const T*)expT, 1, Q, (T*)out, nullptr,
                         base, as_stream(stream));
}

template <typename T>
int dense_backward_launch(int64_t T_, int K, int Q, const void* w, const void* g,
                          const void* expT, void* out, void* stream) {
    BEER_REQUIRE(T_ >= 0 && Q >= 1 && K >= 1 && w && expT && out);
    if (T_ == 0) return BEER_OK;
    // out[t,q] = g_t * sum_k w[t,k] expT[k,q]
    if (sizeof(T) == 4 && big_shape(T_, Q, K) && gemm3_fits(K, 1, Q, 1, 0))
        return gemm3<true, false>(T_, Q, K, (const float*)w, K, 1, (const float*)expT, Q, 1, (float*)out,
                     nullptr, (const float*)g, 0.0, nullptr, 1, 0, as_stream(stream));
    return gemm_plain<T>(T_, Q, K, (const T*)w, K, 1, (const T*)expT, Q, 1, (T*)out, (const T*)g,
                         0.0, as_stream(stream));
}

template <typename T>
int dense_accumulate_launch(int64_t T_, int K, int Q, int G, const void* w, const void* sr,
                            const void* stats, double* acc, void* stream) {
    BEER_REQUIRE(T_ >= 0 && Q >= 1 && K >= 1 && G >= 1 && K % G == 0 && w && stats && acc);
    if (T_ == 0) return BEER_OK;
    // acc[k,q] += sum_t w[t,k] stats[t,q]; frames split over grid.z, fp64 atomics
    if (sizeof(T) == 4 && big_shape(T_, Q, K) && gemm3_fits(1, K, Q, 1, K / G))
        return gemm3<false, false>(K, Q, T_, (const float*)w, 1, K, (const float*)stats, Q, 1, nullptr, acc,
                     nullptr, 0.0, (const float*)sr, G, K / G, as_stream(stream));
    const int gx = (K + TM - 1) / TM, gy = (Q + TN - 1) / TN;
    int64_t gz = (2048 + (int64_t)gx * gy - 1) / ((int64_t)gx * gy);
    const int64_t max_z = (T_ + 255) / 256;
    if (gz > max_z) gz = max_z;
    if (gz < 1) gz = 1;
    int64_t kpb = (T_ + gz - 1) / gz;
    kpb = (kpb + TK - 1) / TK * TK;
    gz = (T_ + kpb - 1) / kpb;
    hipLaunchKernelGGL((gemm_kernel<T, double, true>), dim3(gx, gy, (unsigned)gz), dim3(256), 0,
                       as_stream(stream), (int64_t)K, Q, T_, (const T*)w, (int64_t)1, (int64_t)K,
                       (const T*)stats, (int64_t)Q, (int64_t)1, acc, (const T*)nullptr, 0.0, kpb,
                       (const T*)sr, G, K / G);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

template <typename T>
int rowdot_launch(int64_t T_, int K, const void* a, const void* b, void* out, void* stream) {
    BEER_REQUIRE(T_ >= 0 && K >= 1);
    if (T_ == 0) return BEER_OK;
    hipLaunchKernelGGL(rowdot_kernel<T>, dim3((unsigned)((T_ + 3) / 4)), dim3(256), 0,
                       as_stream(stream), T_, K, (const T*)a, (const T*)b, (T*)out);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

template <typename T>
int softmax_groups_launch(int64_t T_, int S, int G, const void* pc, const void* logw,
                          void* log_norm, void* resps, void* stream) {
    BEER_REQUIRE(T_ >= 0 && S >= 1 && G >= 1 && pc);
    if (T_ == 0) return BEER_OK;
    const int64_t n = T_ * S;
    hipLaunchKernelGGL(softmax_groups_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       as_stream(stream), T_, S, G, (const T*)pc, (const T*)logw, (T*)log_norm,
                       (T*)resps);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

template <typename T>
int suffstats_mean_launch(int cov, int64_t T_, int ns, int D, const void* X, void* out,
                          void* stream) {
    BEER_REQUIRE(T_ >= 0 && ns >= 1 && D >= 1 && cov >= 0 && cov <= 2);
    if (T_ == 0) return BEER_OK;
    const int64_t total = T_ * stats_dim(cov, D);
    const int blocks = (int)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
    hipLaunchKernelGGL(suffstats_mean_kernel<T>, dim3(blocks), dim3(256), 0, as_stream(stream),
                       cov, T_, ns, D, (const T*)X, (T*)out);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

template <typename T>
int suffstats_backward_launch(int cov, int64_t T_, int ns, int D, const void* X, const void* gs,
                              void* gx, void* stream) {
    BEER_REQUIRE(T_ >= 0 && ns >= 1 && D >= 1 && cov >= 0 && cov <= 2);
    if (T_ == 0) return BEER_OK;
    if (cov == BEER_FULL && D <= 64 && D >= 8) {
        const size_t lds = 4 * ((size_t)D * (D + 1) + 2 * D) * sizeof(T);
        const int64_t wgs = (T_ + 3) / 4;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(suffstats_backward_full_kernel<T>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, beer::kMaxDynLds);
        hipLaunchKernelGGL(suffstats_backward_full_kernel<T>,
                           dim3((unsigned)(wgs > 65536 ? 65536 : wgs)), dim3(256), lds,
                           as_stream(stream), T_, ns, D, (const T*)X, (const T*)gs, (T*)gx);
        BEER_LAUNCH_CHECK();
        return BEER_OK;
    }
    const int64_t total = T_ * ns * D;
    const int blocks = (int)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
    hipLaunchKernelGGL(suffstats_backward_kernel<T>, dim3(blocks), dim3(256), 0, as_stream(stream),
                       cov, T_, ns, D, (const T*)X, (const T*)gs, (T*)gx);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

}  // namespace

extern "C" {

int beer_dense_llh(int dtype, int64_t T, int Q, int K, const void* stats, const void* exp_stats,
                   double base, void* out, void* stream) {
    BEER_DISPATCH(dtype, dense_llh_launch, T, Q, K, stats, exp_stats, base, out, stream);
}

int beer_dense_llh_backward(int dtype, int64_t T, int K, int Q, const void* weights,
                            const void* grad, const void* exp_stats, void* out, void* stream) {
    BEER_DISPATCH(dtype, dense_backward_launch, T, K, Q, weights, grad, exp_stats, out, stream);
}

int beer_dense_accumulate(int dtype, int64_t T, int K, int Q, int G, const void* weights,
                          const void* state_resps, const void* stats, double* acc,
                          void* stream) {
    BEER_DISPATCH(dtype, dense_accumulate_launch, T, K, Q, G, weights, state_resps, stats, acc,
                  stream);
}

int beer_rowdot(int dtype, int64_t T, int K, const void* a, const void* b, void* out,
                void* stream) {
    BEER_DISPATCH(dtype, rowdot_launch, T, K, a, b, out, stream);
}

int beer_softmax_groups(int dtype, int64_t T, int S, int G, const void* pc_llh,
                        const void* log_weights, void* log_norm, void* resps, void* stream) {
    BEER_DISPATCH(dtype, softmax_groups_launch, T, S, G, pc_llh, log_weights, log_norm, resps,
                  stream);
}

int beer_suffstats_mean(int dtype, int cov, int64_t T, int ns, int D, const void* X, void* out,
                        void* stream) {
    BEER_DISPATCH(dtype, suffstats_mean_launch, cov, T, ns, D, X, out, stream);
}

int beer_suffstats_backward(int dtype, int cov, int64_t T, int ns, int D, const void* X,
                            const void* grad_stats, void* grad_X, void* stream) {
    BEER_DISPATCH(dtype, suffstats_backward_launch, cov, T, ns, D, X, grad_stats, grad_X, stream);
}

}  // extern "C"
