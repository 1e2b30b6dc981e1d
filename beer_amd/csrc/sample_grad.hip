// Gradient of a prior's expected log-likelihood w.r.t. the FRAMES it was evaluated on:
//
//     out[t, :] = grad[t] * sum_k weights[t, k] * d l_k(x_t) / d x_t,      l_k(x) = phi(x) . E[T]_k
//
// This is what the VAE needs from its prior when it draws ONE sample z_t per frame
// (beer/models/vae.py:63-86): the "statistics" it hands over are then phi(z_t) itself, the
// prior's E-step and accumulation are the frame kernels (estep_bf16.hip), and the chain
// `(pc_llhs * resps).sum(-1)` -> statistics -> samples that the reference's autograd walks
// (mixture.py:79,92; hmm.py:81-87; normalwishart.py:30-38) collapses into the expression
// above -- no [T, Q] operand anywhere (Q = D^2 + D + 2 = 4162 at D = 64: 16.6 GB per million
// frames for the statistics and again for their gradient in dense.hip's route).
//
// With E[T]_k = [E1_k (D) | E2_k (D x D, or D, or 1) | c, c'] and phi(x) = [x, -x x^T / 2, ..]:
//     full       d l_k / d x = E1_k - (E2_k + E2_k^T) x / 2
//     diagonal   d l_k / d x = E1_k - E2_k * x            (elementwise)
//     isotropic  d l_k / d x = E1_k - E2_k x              (E2_k a scalar)
//
// Full covariance, float32, D <= 64 (`sgrad_kernel`): y_k(t) = E1_k + B_k z_t with
// B_k = -(E2_k + E2_k^T) / 2 is a [D x D] x [D x frames] product per component on the matrix
// cores in the E-step's bf16x3 arithmetic (both operands exactly as three bf16 pieces, six
// MFMAs per product, float32 accumulation from C = E1_k), and out += w_tk y_k(t) is four
// float32 multiply-adds per MFMA tile on the vector ALU -- nothing is split inside the loop:
// the frames' pieces are built once per wave and stay in registers for all K components, the
// parameters' pieces are prepared once per call (`sgrad_image_kernel`) as ready-made MFMA
// fragments and copied global -> LDS by the DMA path, component k + 1 in flight while k is
// multiplied.  Algorithmic work: 2 T K D (D + 1) flop (983 GFLOP per million frames at
// K = 120, D = 64), six bf16 MFMAs per float32 product.
// Diagonal / isotropic, float32, D <= 128: `sgrad_diag_kernel` below (one [T, K] x [K, 2 D] product
// on the matrix cores, HBM-bound).
// Everything else (float64, full covariance beyond 64 dimensions): the two steps of dense.hip's route --
// beer_dense_llh_backward, then beer_suffstats_backward -- over chunks of frames whose [frames, Q]
// gradient fits a bounded workspace (256 MiB); small inputs, or no workspace:
// `sgrad_generic_kernel`, a thread per output, float64 accumulation.

#include "common.h"
#include "estep_tiles.h"

using namespace beer;

namespace {

typedef unsigned int sgu4 __attribute__((ext_vector_type(4)));
typedef __bf16 sgbf8 __attribute__((ext_vector_type(8)));
using beer_mfma::f32x4;

#ifndef BEER_SG_WAVES
#define BEER_SG_WAVES 8
#endif
#ifndef BEER_SG_WM
#define BEER_SG_WM 4
#endif
#ifndef BEER_SG_KPS
#define BEER_SG_KPS 1                  // components per stage (one barrier per stage)
#endif
#ifndef BEER_SG_ABL
#define BEER_SG_ABL 0    // timing experiments (wrong results), bits: 1 no barrier / re-staging after
#endif                   // the first component, 2 B fragments read from LDS once per component
constexpr int kSgWaves = BEER_SG_WAVES;   // waves per workgroup (two per SIMD)
constexpr int kSgWM = BEER_SG_WM;         // 16-frame tiles per wave
constexpr int kSgKps = BEER_SG_KPS;
constexpr int kSgFrames = kSgWaves * kSgWM * 16;
constexpr int64_t kSgMinFrames = 4096; // below: the generic kernel

// bytes of one component in the parameter image: [col tile j][k-block kb][piece q] fragments of
// 1 KiB (lane-linear: 16 bytes per lane), then E1_k as NJ * 16 floats padded to 1 KiB
__host__ __device__ constexpr int sg_chunks(int NKB) { return 2 * NKB * NKB * 3 + 1; }

template <int NKB>
__global__ __launch_bounds__(256) void sgrad_image_kernel(int K, int D, const float* __restrict__ E,
                                                          char* __restrict__ img) {
    constexpr int NJ = 2 * NKB, NCH = sg_chunks(NKB);
    const int Q = D * D + D + 2;
    const int64_t total = (int64_t)K * NCH * 64;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * 256) {
        const int lane = (int)(idx & 63);
        const int ch = (int)((idx >> 6) % NCH), k = (int)((idx >> 6) / NCH);
        const float* Ek = E + (size_t)k * Q;
        char* dst = img + ((size_t)k * NCH + ch) * 1024 + lane * 16;
        if (ch == NCH - 1) {
            // E1_k: floats 4 lane .. 4 lane + 3 of the padded row
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int d = lane * 4 + e;
                v[e] = (d < D && d < NJ * 16) ? Ek[d] : 0.f;
            }
            *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
            continue;
        }
        const int q = ch % 3, kb = (ch / 3) % NKB, j = ch / (3 * NKB);
        const int fi = lane & 15, fg = lane >> 4, d = j * 16 + fi;
        const int dp0 = kb * 32 + fg * 8;
        unsigned w[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            unsigned short h[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int dp = dp0 + 2 * p + s;
                float v = 0.f;
                if (d < D && dp < D) v = -0.5f * (Ek[D + dp * D + d] + Ek[D + d * D + dp]);
                unsigned short pc[3];
                beer_mfma::split3_scalar(v, pc);
                h[s] = pc[q];
            }
            w[p] = (unsigned)h[0] | ((unsigned)h[1] << 16);
        }
        *reinterpret_cast<sgu4*>(dst) = sgu4{w[0], w[1], w[2], w[3]};
    }
}

template <int NKB>
__global__ __launch_bounds__(kSgWaves * 64) void sgrad_kernel(
    int64_t T_, int D, int K, const float* __restrict__ X, const float* __restrict__ W,
    const float* __restrict__ g, const char* __restrict__ img, float* __restrict__ out) {
    constexpr int NJ = 2 * NKB, NCH = sg_chunks(NKB), WM = kSgWM;
    extern __shared__ __attribute__((aligned(16))) char smem[];          // 2 stages
    static_assert(4 % kSgKps == 0, "stages of 1, 2 or 4 components");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 15, fg = lane >> 4;
    const int64_t t0 = (int64_t)blockIdx.x * kSgFrames + wave * (WM * 16);

    // the frames of this wave as MFMA fragments, three bf16 pieces each: lane (fi, fg) holds
    // dimensions kb * 32 + 8 fg .. + 7 of frame 16 i + fi
    sgu4 zf[WM][NKB][3];
    const float* wrow[WM];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int64_t t = t0 + i * 16 + fi;
        const int64_t tc = t < T_ ? t : T_ - 1;
        wrow[i] = W + tc * K;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int dp = kb * 32 + fg * 8 + e;
                v[e] = (t < T_ && dp < D) ? X[tc * D + dp] : 0.f;
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                unsigned pc[3];
                beer_mfma::split3(v[2 * p], v[2 * p + 1], pc);
#pragma unroll
                for (int q = 0; q < 3; ++q) zf[i][kb][q][p] = pc[q];
            }
        }
    }

    typedef __attribute__((address_space(3))) void* lds_ptr;
    // components k .. k + kSgKps - 1 (those below K) into buffer `buf`
    auto stage = [&](int k, int buf) {
        const char* src = img + (size_t)k * (NCH * 1024) + lane * 16;
        char* dst = smem + buf * (kSgKps * NCH * 1024);
        const int nch = (K - k < kSgKps ? K - k : kSgKps) * NCH;
#pragma unroll
        for (int c = 0; c < (kSgKps * NCH + kSgWaves - 1) / kSgWaves; ++c) {
            const int ch = wave + c * kSgWaves;
            if (ch < nch)
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const sgu4*>(src + ch * 1024),
                                                 (lds_ptr)(dst + ch * 1024), 16, 0, 0);
        }
    };
    auto load_w = [&](int k0, float (&dst)[WM][4]) {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) dst[i][c] = k0 + c < K ? wrow[i][k0 + c] : 0.f;
    };

    f32x4 acc[WM][NJ];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    float wcur[WM][4], wnext[WM][4];
    stage(0, 0);
    if ((BEER_SG_ABL & 1) && K > kSgKps) {
        stage(kSgKps, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    load_w(0, wnext);
    for (int k0 = 0; k0 < K; k0 += 4) {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) wcur[i][c] = wnext[i][c];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int k = k0 + c;
            if (k >= K) break;
            if ((BEER_SG_ABL & 1) && k > 0) {
                // (no barrier, no staging -- but the same fence for the compiler)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            } else if (c % kSgKps == 0) {
                // this stage has landed (this wave's share; the barrier: everybody's), and
                // everybody is done with the previous one, whose buffer the next goes into
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (k + kSgKps < K) stage(k + kSgKps, (k / kSgKps + 1) & 1);
            }
            if (c == 0) load_w(k0 + 4, wnext);
            const char* buf = smem + ((k / kSgKps) & 1) * (kSgKps * NCH * 1024) +
                              (c % kSgKps) * (NCH * 1024);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                sgu4 bfr[NKB][3];
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        bfr[kb][q] = *reinterpret_cast<const sgu4*>(
                            buf + ((((BEER_SG_ABL & 2) ? 0 : j) * NKB + kb) * 3 + q) * 1024 + lane * 16);
                // C = E1_k: a lane holds y[dimension 16 j + 4 fg + e][frame fi]
                const f32x4 c0 = *reinterpret_cast<const f32x4*>(
                    buf + (NCH - 1) * 1024 + (j * 16 + fg * 4) * 4);
                f32x4 y[WM];
#pragma unroll
                for (int i = 0; i < WM; ++i) y[i] = c0;
                // smallest products first
                constexpr int PZ[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
                for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                        for (int i = 0; i < WM; ++i)
                            y[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                __builtin_bit_cast(sgbf8, bfr[kb][PB[pr]]),
                                __builtin_bit_cast(sgbf8, zf[i][kb][PZ[pr]]), y[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[i][j][e] = __builtin_fmaf(wcur[i][c], y[i][e], acc[i][j][e]);
            }
        }
    }

#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int64_t t = t0 + i * 16 + fi;
        if (t >= T_) continue;
        const float sc = g ? g[t] : 1.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int d0 = j * 16 + fg * 4;
            if (d0 + 3 < D && (D & 3) == 0) {
                *reinterpret_cast<f32x4*>(out + t * D + d0) =
                    f32x4{sc * acc[i][j][0], sc * acc[i][j][1], sc * acc[i][j][2],
                          sc * acc[i][j][3]};
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (d0 + e < D) out[t * D + d0 + e] = sc * acc[i][j][e];
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Diagonal / isotropic covariance, float32, D <= 128 (`sgrad_diag_kernel`):
//     out[t, d] = grad[t] * (S1[t, d] - x[t, d] S2[t, d]),   [S1 | S2] = w [T, K] x [E1 | E2] [K, 2 D]
// one product on the matrix cores in the same bf16x3 arithmetic -- the posteriors as the frame-side
// operand, split on the fly (8 values per lane and 32 components), [E1 | E2] prepared once per
// call as MFMA fragments (`sgrad_diag_image_kernel`; isotropic: E2_k repeated over the
// dimensions) and staged through LDS 32 components at a time -- and the combination with the
// frame in the epilogue: the lane that holds S1[t, d] also holds S2[t, d].  The call streams
// w, x and out once (4 (K + 2 D) bytes per frame): HBM-bound.
// ---------------------------------------------------------------------------
constexpr int kSdWaves = 8, kSdWM = 2, kSdFrames = kSdWaves * kSdWM * 16;

inline int sd_tiles(int D) { return D <= 32 ? 2 : (D <= 48 ? 3 : (D <= 64 ? 4 : 8)); }   // of 16 dims

template <int NJ2>
__global__ __launch_bounds__(256) void sgrad_diag_image_kernel(int cov, int K, int D,
                                                               const float* __restrict__ E,
                                                               char* __restrict__ img) {
    constexpr int NJ = 2 * NJ2, DP = NJ2 * 16;
    const int Q = stats_dim(cov, D), KB = (K + 31) / 32;
    const int64_t total = (int64_t)KB * NJ * 3 * 64;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * 256) {
        const int lane = (int)(idx & 63);
        const int ch = (int)(idx >> 6), q = ch % 3, j = (ch / 3) % NJ, kb = ch / (3 * NJ);
        const int fi = lane & 15, fg = lane >> 4, n = j * 16 + fi;
        const int d = n < DP ? n : n - DP;
        unsigned w[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            unsigned short h[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int k = kb * 32 + fg * 8 + 2 * p + s2;
                float v = 0.f;
                if (k < K && d < D)
                    v = n < DP ? E[(size_t)k * Q + d]
                               : E[(size_t)k * Q + D + (cov == BEER_DIAG ? d : 0)];
                unsigned short pc[3];
                beer_mfma::split3_scalar(v, pc);
                h[s2] = pc[q];
            }
            w[p] = (unsigned)h[0] | ((unsigned)h[1] << 16);
        }
        *reinterpret_cast<sgu4*>(img + (size_t)ch * 1024 + lane * 16) = sgu4{w[0], w[1], w[2], w[3]};
    }
}

template <int NJ2>
__global__ __launch_bounds__(kSdWaves * 64) void sgrad_diag_kernel(
    int64_t T_, int D, int K, const float* __restrict__ X, const float* __restrict__ W,
    const float* __restrict__ g, const char* __restrict__ img, float* __restrict__ out) {
    constexpr int NJ = 2 * NJ2, WM = kSdWM, STAGE = NJ * 3 * 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];          // 2 stages
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 15, fg = lane >> 4, KB = (K + 31) / 32;
    const int64_t t0 = (int64_t)blockIdx.x * kSdFrames + wave * (WM * 16);
    const float* wrow[WM];
    bool live[WM];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int64_t t = t0 + i * 16 + fi;
        live[i] = t < T_;
        wrow[i] = W + (live[i] ? t : T_ - 1) * K;
    }
    typedef __attribute__((address_space(3))) void* lds_ptr;
    auto stage = [&](int kb, int buf) {
        const char* src = img + (size_t)kb * STAGE + lane * 16;
        char* dst = smem + buf * STAGE;
#pragma unroll
        for (int c = 0; c < (NJ * 3 + kSdWaves - 1) / kSdWaves; ++c) {
            const int ch = wave + c * kSdWaves;
            if (ch < NJ * 3)
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const sgu4*>(src + ch * 1024),
                                                 (lds_ptr)(dst + ch * 1024), 16, 0, 0);
        }
    };
    auto load_w = [&](int kb, float (&dst)[WM][8]) {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = kb * 32 + fg * 8 + e;
                dst[i][e] = (k < K && live[i]) ? wrow[i][k] : 0.f;
            }
    };
    f32x4 acc[WM][NJ];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float wnext[WM][8];
    stage(0, 0);
    load_w(0, wnext);
    for (int kb = 0; kb < KB; ++kb) {
        sgu4 af[WM][3];
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                unsigned pc[3];
                beer_mfma::split3(wnext[i][2 * p], wnext[i][2 * p + 1], pc);
#pragma unroll
                for (int q = 0; q < 3; ++q) af[i][q][p] = pc[q];
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kb + 1 < KB) {
            stage(kb + 1, (kb + 1) & 1);
            load_w(kb + 1, wnext);
        }
        const char* buf = smem + (kb & 1) * STAGE;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            sgu4 bfr[3];
#pragma unroll
            for (int q = 0; q < 3; ++q)
                bfr[q] = *reinterpret_cast<const sgu4*>(buf + (j * 3 + q) * 1024 + lane * 16);
            constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
            for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                for (int i = 0; i < WM; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(sgbf8, bfr[PB[pr]]), __builtin_bit_cast(sgbf8, af[i][PA[pr]]),
                        acc[i][j], 0, 0, 0);
        }
    }
    // a lane holds [S1 | S2][frame fi][column 16 j + 4 fg + e]: S1 in tiles j < NJ2, S2 of the
    // same dimensions in tiles j + NJ2
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int64_t t = t0 + i * 16 + fi;
        if (t >= T_) continue;
        const float sc = g ? g[t] : 1.f;
#pragma unroll
        for (int j = 0; j < NJ2; ++j) {
            const int d0 = j * 16 + fg * 4;
            if (d0 + 3 < D && (D & 3) == 0) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(X + t * D + d0);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = sc * (acc[i][j][e] - x[e] * acc[i][j + NJ2][e]);
                *reinterpret_cast<f32x4*>(out + t * D + d0) = v;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (d0 + e < D)
                        out[t * D + d0 + e] =
                            sc * (acc[i][j][e] - X[t * D + d0 + e] * acc[i][j + NJ2][e]);
            }
        }
    }
}

// any dtype / covariance type / dimension: a thread per output, float64 accumulation
template <typename T>
__global__ __launch_bounds__(256) void sgrad_generic_kernel(
    int cov, int64_t T_, int D, int K, const T* __restrict__ X, const T* __restrict__ W,
    const T* __restrict__ g, const T* __restrict__ E, T* __restrict__ out) {
    const int Q = stats_dim(cov, D);
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < T_ * D;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = idx / D;
        const int d = (int)(idx % D);
        const T* x = X + t * D;
        const T* w = W + t * K;
        double s = 0.0;
        for (int k = 0; k < K; ++k) {
            const double wk = (double)w[k];
            if (wk == 0.0) continue;
            const T* Ek = E + (size_t)k * Q;
            double v = (double)Ek[d];
            if (cov == BEER_FULL) {
                double r = 0.0;
                for (int j = 0; j < D; ++j)
                    r += ((double)Ek[D + d * D + j] + (double)Ek[D + j * D + d]) * (double)x[j];
                v -= 0.5 * r;
            } else if (cov == BEER_DIAG) {
                v -= (double)Ek[D + d] * (double)x[d];
            } else {
                v -= (double)Ek[D] * (double)x[d];
            }
            s += wk * v;
        }
        out[idx] = (T)((g ? (double)g[t] : 1.0) * s);
    }
}

inline int sg_blocks(int D) { return D <= 32 ? 1 : 2; }

inline bool sg_fast(int dtype, int cov, int64_t T_, int D, int K) {
    return dtype == BEER_F32 && cov == BEER_FULL && D >= 8 && D <= 64 && K >= 1 &&
           T_ >= kSgMinFrames;
}

inline size_t sg_image_bytes(int D, int K) {
    return (size_t)K * sg_chunks(sg_blocks(D)) * 1024;
}

inline bool sd_fast(int dtype, int cov, int64_t T_, int D, int K) {
    return dtype == BEER_F32 && (cov == BEER_DIAG || cov == BEER_ISO) && D >= 1 && D <= 128 &&
           K >= 1 && T_ >= kSgMinFrames;
}
inline size_t sd_image_bytes(int D, int K) {
    return (size_t)((K + 31) / 32) * 2 * sd_tiles(D) * 3 * 1024;
}

// the chunked two-step route: frames per chunk, bytes of its [frames, Q] gradient
constexpr size_t kSgChunkBytes = (size_t)256 << 20;
inline int64_t sg_chunk_frames(int dtype, int cov, int64_t T_, int D) {
    const size_t row = (size_t)stats_dim(cov, D) * (dtype == BEER_F64 ? 8 : 4);
    int64_t n = (int64_t)(kSgChunkBytes / row);
    if (n < 1024) n = 1024;
    return n < T_ ? n : T_;
}
inline size_t sg_chunk_bytes(int dtype, int cov, int64_t T_, int D) {
    return (size_t)sg_chunk_frames(dtype, cov, T_, D) * stats_dim(cov, D) *
           (dtype == BEER_F64 ? 8 : 4);
}

template <typename T>
int sgrad_generic_launch(int cov, int64_t T_, int D, int K, const void* X, const void* W,
                         const void* g, const void* E, void* out, void* stream) {
    const int64_t total = T_ * D;
    const int blocks = (int)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
    hipLaunchKernelGGL(sgrad_generic_kernel<T>, dim3(blocks), dim3(256), 0, as_stream(stream), cov,
                       T_, D, K, (const T*)X, (const T*)W, (const T*)g, (const T*)E, (T*)out);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

template <int NJ2>
int sgrad_diag_launch(int cov, int64_t T_, int D, int K, const float* X, const float* W,
                      const float* g, const float* E, float* out, char* img, hipStream_t s) {
    const int64_t items = (int64_t)((K + 31) / 32) * 2 * NJ2 * 3 * 64;
    hipLaunchKernelGGL(sgrad_diag_image_kernel<NJ2>, dim3((unsigned)((items + 255) / 256)),
                       dim3(256), 0, s, cov, K, D, E, img);
    BEER_LAUNCH_CHECK();
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sgrad_diag_kernel<NJ2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, beer::kMaxDynLds);
    hipLaunchKernelGGL(sgrad_diag_kernel<NJ2>, dim3((unsigned)((T_ + kSdFrames - 1) / kSdFrames)),
                       dim3(kSdWaves * 64), 2 * 2 * NJ2 * 3 * 1024, s, T_, D, K, X, W, g, img, out);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

template <int NKB>
int sgrad_fast_launch(int64_t T_, int D, int K, const float* X, const float* W, const float* g,
                      const float* E, float* out, char* img, hipStream_t s) {
    const int64_t items = (int64_t)K * sg_chunks(NKB) * 64;
    hipLaunchKernelGGL(sgrad_image_kernel<NKB>, dim3((unsigned)((items + 255) / 256)), dim3(256), 0,
                       s, K, D, E, img);
    BEER_LAUNCH_CHECK();
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sgrad_kernel<NKB>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, beer::kMaxDynLds);
    hipLaunchKernelGGL(sgrad_kernel<NKB>, dim3((unsigned)((T_ + kSgFrames - 1) / kSgFrames)),
                       dim3(kSgWaves * 64), 2 * kSgKps * sg_chunks(NKB) * 1024, s, T_, D, K, X, W, g,
                       img, out);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

}  // namespace

extern "C" {

size_t beer_frames_llh_backward_workspace_bytes(int dtype, int cov, int64_t T, int D, int K) {
    if (T < kSgMinFrames || D < 1 || K < 1 || cov < 0 || cov > 2) return 0;
    if (dtype != BEER_F32 && dtype != BEER_F64) return 0;
    if (sg_fast(dtype, cov, T, D, K)) return sg_image_bytes(D, K);
    if (sd_fast(dtype, cov, T, D, K)) return sd_image_bytes(D, K);
    return sg_chunk_bytes(dtype, cov, T, D);
}

int beer_frames_llh_backward(int dtype, int cov, int64_t T, int D, int K, const void* X,
                             const void* weights, const void* grad, const void* exp_stats,
                             void* out, void* workspace, size_t workspace_bytes, void* stream) {
    BEER_REQUIRE(T >= 0 && D >= 1 && K >= 1 && cov >= 0 && cov <= 2);
    BEER_REQUIRE(dtype == BEER_F32 || dtype == BEER_F64);
    if (T == 0) return BEER_OK;
    BEER_REQUIRE(X && weights && exp_stats && out);
    if (sg_fast(dtype, cov, T, D, K) && workspace && workspace_bytes >= sg_image_bytes(D, K)) {
        if (sg_blocks(D) == 1)
            return sgrad_fast_launch<1>(T, D, K, (const float*)X, (const float*)weights,
                                        (const float*)grad, (const float*)exp_stats, (float*)out,
                                        (char*)workspace, as_stream(stream));
        return sgrad_fast_launch<2>(T, D, K, (const float*)X, (const float*)weights,
                                    (const float*)grad, (const float*)exp_stats, (float*)out,
                                    (char*)workspace, as_stream(stream));
    }
    if (sd_fast(dtype, cov, T, D, K) && workspace && workspace_bytes >= sd_image_bytes(D, K)) {
#define BEER_SD(NJ2_)                                                                            \
    return sgrad_diag_launch<NJ2_>(cov, T, D, K, (const float*)X, (const float*)weights,         \
                                   (const float*)grad, (const float*)exp_stats, (float*)out,     \
                                   (char*)workspace, as_stream(stream))
        switch (sd_tiles(D)) {
            case 2: BEER_SD(2);
            case 3: BEER_SD(3);
            case 4: BEER_SD(4);
            default: BEER_SD(8);
        }
#undef BEER_SD
    }
    if (T >= kSgMinFrames && workspace && workspace_bytes >= sg_chunk_bytes(dtype, cov, T, D)) {
        const int64_t chunk = sg_chunk_frames(dtype, cov, T, D);
        const size_t elem = dtype == BEER_F64 ? 8 : 4;
        const int Q = stats_dim(cov, D);
        for (int64_t c0 = 0; c0 < T; c0 += chunk) {
            const int64_t n = T - c0 < chunk ? T - c0 : chunk;
            int rc = beer_dense_llh_backward(
                dtype, n, K, Q, (const char*)weights + (size_t)c0 * K * elem,
                grad ? (const char*)grad + (size_t)c0 * elem : nullptr, exp_stats, workspace, stream);
            if (rc != BEER_OK) return rc;
            rc = beer_suffstats_backward(dtype, cov, n, 1, D, (const char*)X + (size_t)c0 * D * elem,
                                         workspace, (char*)out + (size_t)c0 * D * elem, stream);
            if (rc != BEER_OK) return rc;
        }
        return BEER_OK;
    }
    BEER_DISPATCH(dtype, sgrad_generic_launch, cov, T, D, K, X, weights, grad, exp_stats, out,
                  stream);
}

}  // extern "C"
