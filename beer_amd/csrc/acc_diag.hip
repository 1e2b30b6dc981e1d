// Weighted statistics of diagonal / isotropic Gaussians from float32 weights in memory
// (`accd_kernel`):
//     acc[k, :] += sum_t w[t, k] phi(x_t),   phi(x) = [x, -x^2 / 2, -1/2, 1/2]     (diagonal)
//                                                     [x, -|x|^2 / 2, -1/2, D/2]   (isotropic)
// NormalSet.accumulate (beer/models/normalset.py:121-123) with the responsibilities an HMM
// hands a set of single Gaussians -- its state posteriors [T, S] (hmm.py:94-100) -- or a mixture
// its [T, K] responsibilities (mixture.py:95-102).  The caller that matters is the prior of a
// VAE (config 4: S = 120 states over a 64-dimensional latent space, one million samples per
// minibatch): 2 T K (2 D) flop against 4 (K + D) bytes per frame, 31 GFLOP against 0.74 GB --
// HBM-bound at the matrix cores' rate, VALU-bound on the exact float32 kernels (0.85 ms).
//
// One product [2 D, T] x [T, K] in the E-step's bf16x3 arithmetic (both operands exactly as three
// bf16 pieces, six v_mfma_f32_16x16x32_bf16 per product, float32 sums over at most kAdChain
// frames, fp64 atomics beyond).  A workgroup of four waves walks kAdChain frames in tiles of
// 64: the weights and the frames of a tile are fetched row-major into registers a tile ahead and
// parked TRANSPOSED in LDS ([component][68], [dimension][68]: 8 consecutive frames of a row are
// one MFMA operand), a wave owns CT component tiles x all 2 NJ2 statistic tiles, splits its
// operands on the fly (x and x^2 of a dimension from the same read) and sums the counts N_k on
// the vector ALU from the very weights it multiplies.

#include "common.h"
#include "estep_mfma.h"
#include "estep_tiles.h"

using namespace beer;

namespace {

using beer_mfma::f32x4;
typedef unsigned int adu4 __attribute__((ext_vector_type(4)));
typedef __bf16 adbf8 __attribute__((ext_vector_type(8)));

constexpr int kAdFT = 64;                 // frames per tile
constexpr int kAdLD = 68;                 // floats per transposed row (16-byte aligned, 4 of padding)
#ifndef BEER_AD_CHAIN
#define BEER_AD_CHAIN 2048
#endif
constexpr int kAdChain = BEER_AD_CHAIN;   // frames a workgroup sums in float32 ...
#ifndef BEER_AD_INNER
#define BEER_AD_INNER 512
#endif
// ... of which the matrix cores' accumulators sum kAdInner (16 k-steps of six MFMAs: the MFMA
// truncates its addends at ulp(C)/32, a long chain over products of one sign drifts low --
// DESIGN.md 5.1b) before the vector ALU adds them, rounding to nearest, into a second set
constexpr int kAdInner = BEER_AD_INNER;
static_assert(kAdChain % kAdInner == 0 && kAdInner % kAdFT == 0, "whole tiles per inner chain");
constexpr int64_t kAdMinFrames = 16384;   // below: the exact kernels (launch-bound there)
constexpr size_t kAdMaxPartialBytes = (size_t)512 << 20;   // largest partial-sum workspace asked for

template <int NJ2, int CT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void accd_kernel(
    int cov, int64_t T_, int D, int K, const float* __restrict__ X, const float* __restrict__ W,
    double* __restrict__ acc, float* __restrict__ part, float* __restrict__ cpart) {
    constexpr int KB = 64 * CT;           // components per workgroup
    constexpr int DP = 16 * NJ2;          // dimensions, padded
    constexpr int LW = DP <= 16 ? 16 : (DP <= 32 ? 32 : 64);     // lanes along a frame's dimensions
    constexpr int NW = kAdFT * KB / 256, NZ = kAdFT * LW / 256;
    constexpr int FW = 256 / KB, FZ = 256 / LW;                  // frames one pass of 256 lanes covers
    __shared__ __attribute__((aligned(16))) float Wt[KB * kAdLD];
    __shared__ __attribute__((aligned(16))) float Zt[DP * kAdLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fi = lane & 15, fg = lane >> 4;
    const int kb0 = blockIdx.y * KB;
    const int64_t tb = (int64_t)blockIdx.x * kAdChain;
    const int nfr = (int)(tb + kAdChain < T_ ? kAdChain : T_ - tb);       // frames of this chain

    // The chain's rows as buffer resources: a lane's offset within a pass is fixed (one register),
    // the pass and the tile add a uniform offset (in the vector offset: the range check does not
    // see the scalar one), rows past the chain's end read as 0.
    constexpr int kOob = 0x40000000;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(W + tb * K), 0, nfr * K * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(X + tb * D), 0, nfr * D * 4, 0x00020000);
    const int wc = tid % KB, wt = tid / KB, zd = tid % LW, zt = tid / LW;
    const int vo_w = kb0 + wc < K ? (wt * K + kb0 + wc) * 4 : kOob;
    const int vo_z = zd < D ? (zt * D + zd) * 4 : kOob;
    float wreg[NW], zreg[NZ];
    auto fetch = [&](int f0) {                                   // tile at frame f0 of the chain
#pragma unroll
        for (int n = 0; n < NW; ++n)
            wreg[n] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                    rw, vo_w + (f0 + n * FW) * K * 4, 0, 0));
#pragma unroll
        for (int n = 0; n < NZ; ++n)
            zreg[n] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                    rx, vo_z + (f0 + n * FZ) * D * 4, 0, 0));
    };
    auto park = [&]() {
#pragma unroll
        for (int n = 0; n < NW; ++n) Wt[wc * kAdLD + wt + n * FW] = wreg[n];
        if (zd < DP) {
#pragma unroll
            for (int n = 0; n < NZ; ++n) Zt[zd * kAdLD + zt + n * FZ] = zreg[n];
        }
    };
    // 8 consecutive frames of one transposed row -> the three bf16 pieces of an MFMA operand
    auto pieces = [](const f32x4& a, const f32x4& b, adu4 (&f)[3]) {
        unsigned pc[3];
        beer_mfma::split3(a[0], a[1], pc);
#pragma unroll
        for (int q = 0; q < 3; ++q) f[q][0] = pc[q];
        beer_mfma::split3(a[2], a[3], pc);
#pragma unroll
        for (int q = 0; q < 3; ++q) f[q][1] = pc[q];
        beer_mfma::split3(b[0], b[1], pc);
#pragma unroll
        for (int q = 0; q < 3; ++q) f[q][2] = pc[q];
        beer_mfma::split3(b[2], b[3], pc);
#pragma unroll
        for (int q = 0; q < 3; ++q) f[q][3] = pc[q];
    };

    f32x4 sum[CT][2 * NJ2], tot[CT][2 * NJ2];
    float cnt[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        cnt[ct] = 0.f;
#pragma unroll
        for (int j = 0; j < 2 * NJ2; ++j) sum[ct][j] = tot[ct][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    auto fold = [&]() {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int j = 0; j < 2 * NJ2; ++j) {
                tot[ct][j] += sum[ct][j];
                sum[ct][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
    };

    fetch(0);
    for (int f0 = 0; f0 < nfr; f0 += kAdFT) {
        __syncthreads();                  // every wave is done with the tile before
        park();
        __syncthreads();
        if (f0 + kAdFT < nfr) fetch(f0 + kAdFT);
        if (f0 && f0 % kAdInner == 0) fold();
#pragma unroll
        for (int ks = 0; ks < kAdFT / 32; ++ks) {
            adu4 cf[CT][3];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const float* row = Wt + ((wave * CT + ct) * 16 + fi) * kAdLD + ks * 32 + fg * 8;
                const f32x4 a = *reinterpret_cast<const f32x4*>(row);
                const f32x4 b = *reinterpret_cast<const f32x4*>(row + 4);
                cnt[ct] += ((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3]));
                pieces(a, b, cf[ct]);
            }
            // (one statistic tile pair at a time: hipcc otherwise hoists every LDS read and split
            // of the tile above the first MFMA and spills)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NJ2; ++j) {
                const float* row = Zt + (j * 16 + fi) * kAdLD + ks * 32 + fg * 8;
                const f32x4 a = *reinterpret_cast<const f32x4*>(row);
                const f32x4 b = *reinterpret_cast<const f32x4*>(row + 4);
                adu4 xf[3], qf[3];
                pieces(a, b, xf);
                pieces(a * a, b * b, qf);
                // the six leading partial products, smallest first
                constexpr int PS[6] = {2, 1, 0, 1, 0, 0}, PC[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
                for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        sum[ct][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(adbf8, xf[PS[pr]]),
                            __builtin_bit_cast(adbf8, cf[ct][PC[pr]]), sum[ct][j], 0, 0, 0);
                        sum[ct][j + NJ2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(adbf8, qf[PS[pr]]),
                            __builtin_bit_cast(adbf8, cf[ct][PC[pr]]), sum[ct][j + NJ2], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    fold();
    // a lane holds [component fi][statistic 16 j + 4 fg + e]; the counts: its share of the frames
    if (part) {
        // this chain's sums as they are, [chain][component][2 DP] (+ counts [chain][component]):
        // `accd_reduce_kernel` adds the chains up in fp64 -- 8 M fp64 atomics of 489 workgroups
        // onto 15 k addresses cost 0.22 of the 0.56 ms this kernel took at config 4
        const int KP = gridDim.y * KB;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int kl = kb0 + (wave * CT + ct) * 16 + fi;
            float n = cnt[ct];
            n += __shfl_xor(n, 16);
            n += __shfl_xor(n, 32);
            float* row = part + ((size_t)blockIdx.x * KP + kl) * (2 * DP) + 4 * fg;
#pragma unroll
            for (int j = 0; j < 2 * NJ2; ++j) *reinterpret_cast<f32x4*>(row + 16 * j) = tot[ct][j];
            if (fg == 0) cpart[(size_t)blockIdx.x * KP + kl] = n;
        }
        return;
    }
    const int Q = stats_dim(cov, D);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int k = kb0 + (wave * CT + ct) * 16 + fi;
        double n = (double)cnt[ct], sq = 0.0;
#pragma unroll
        for (int j = 0; j < NJ2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (j * 16 + 4 * fg + e < D) sq += (double)tot[ct][j + NJ2][e];
        n += __shfl_xor(n, 16);
        n += __shfl_xor(n, 32);
        if (cov == BEER_ISO) {
            sq += __shfl_xor(sq, 16);
            sq += __shfl_xor(sq, 32);
        }
        if (k >= K) continue;
        double* row = acc + (size_t)k * Q;
#pragma unroll
        for (int j = 0; j < NJ2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int d = j * 16 + 4 * fg + e;
                if (d >= D) continue;
                atomicAdd(row + d, (double)tot[ct][j][e]);
                if (cov == BEER_DIAG) atomicAdd(row + D + d, -0.5 * (double)tot[ct][j + NJ2][e]);
            }
        if (fg == 0) {
            if (cov == BEER_ISO) atomicAdd(row + D, -0.5 * sq);
            atomicAdd(row + Q - 2, -0.5 * n);
            atomicAdd(row + Q - 1, (cov == BEER_ISO ? 0.5 * (double)D : 0.5) * n);
        }
    }
}

// the chains' partial sums -> acc (+=): a thread per (component, column) and slice of the chains
constexpr int kAdSlices = 16;
__global__ __launch_bounds__(256) void accd_reduce_kernel(int cov, int D, int K, int DP, int KP,
                                                          int nchains,
                                                          const float* __restrict__ part,
                                                          const float* __restrict__ cpart,
                                                          double* __restrict__ acc) {
    const int SP = 2 * DP, idx = blockIdx.x * 256 + threadIdx.x;
    const int k = idx / (SP + 1), col = idx - k * (SP + 1);
    if (k >= K) return;
    double v = 0.0;
    if (col < SP) {
        const int d = col < DP ? col : col - DP;
        if (d >= D) return;
        for (int c = blockIdx.y; c < nchains; c += kAdSlices)
            v += (double)part[((size_t)c * KP + k) * SP + col];
    } else {
        for (int c = blockIdx.y; c < nchains; c += kAdSlices) v += (double)cpart[(size_t)c * KP + k];
    }
    const int Q = stats_dim(cov, D);
    double* row = acc + (size_t)k * Q;
    if (col < DP) {
        atomicAdd(row + col, v);
    } else if (col < SP) {
        atomicAdd(row + D + (cov == BEER_DIAG ? col - DP : 0), -0.5 * v);
    } else {
        atomicAdd(row + Q - 2, -0.5 * v);
        atomicAdd(row + Q - 1, (cov == BEER_ISO ? 0.5 * (double)D : 0.5) * v);
    }
}

inline int ad_kp(int K) { return K > 64 ? (K + 127) / 128 * 128 : 64; }
inline int64_t ad_chains(int64_t T_) { return (T_ + kAdChain - 1) / kAdChain; }

template <int NJ2>
int launch_accd(int cov, int64_t T_, int D, int K, const float* X, const float* W, double* acc,
                float* part, float* cpart, hipStream_t s) {
    const unsigned gx = (unsigned)ad_chains(T_);
    if (K > 64)
        hipLaunchKernelGGL((accd_kernel<NJ2, 2>), dim3(gx, (K + 127) / 128), dim3(256), 0, s, cov,
                           T_, D, K, X, W, acc, part, cpart);
    else
        hipLaunchKernelGGL((accd_kernel<NJ2, 1>), dim3(gx, 1), dim3(256), 0, s, cov, T_, D, K, X, W,
                           acc, part, cpart);
    BEER_LAUNCH_CHECK();
    if (part) {
        const int DP = 16 * NJ2;
        hipLaunchKernelGGL(accd_reduce_kernel, dim3((K * (2 * DP + 1) + 255) / 256, kAdSlices),
                           dim3(256), 0, s, cov, D, K, DP, ad_kp(K), (int)gx, part, cpart, acc);
        BEER_LAUNCH_CHECK();
    }
    return BEER_OK;
}

}  // namespace

namespace beer_mfma {

bool supported_acc_diag(int cov, int64_t T_, int D, int K) {
    // (K: the chain's rows are addressed with 32-bit byte offsets, kAdChain K 4 < 2^31)
    return (cov == BEER_DIAG || cov == BEER_ISO) && D >= 1 && D <= 64 && K >= 16 && K <= 65536 &&
           T_ >= kAdMinFrames;
}

// partial sums of every chain of frames: [chains][KP][2 DP] + [chains][KP] floats
size_t acc_diag_workspace_bytes(int cov, int64_t T_, int D, int K) {
    if (!supported_acc_diag(cov, T_, D, K)) return 0;
    const int DP = (D + 15) / 16 * 16;
    const size_t bytes = (size_t)ad_chains(T_) * ad_kp(K) * (2 * DP + 1) * sizeof(float) + 256;
    // (many components x many frames: the partial sums would outgrow what they save -- 32 MB
    // at config 4 -- and the call flushes with atomics instead)
    return bytes <= kAdMaxPartialBytes ? bytes : 0;
}

int acc_diag_bf16x3(int cov, int64_t T_, int D, int K, const float* X, const float* W, double* acc,
                    void* ws, size_t ws_bytes, hipStream_t s) {
    float *part = nullptr, *cpart = nullptr;
    const size_t need = acc_diag_workspace_bytes(cov, T_, D, K);
    // (without the scratch: fp64 atomics straight into `acc` -- same sums, a third slower)
    if (ws && need && ws_bytes >= need && ad_chains(T_) > 1) {
        const int DP = (D + 15) / 16 * 16;
        part = reinterpret_cast<float*>(ws);
        cpart = part + (size_t)ad_chains(T_) * ad_kp(K) * 2 * DP;
    }
    switch ((D + 15) / 16) {
        case 1: return launch_accd<1>(cov, T_, D, K, X, W, acc, part, cpart, s);
        case 2: return launch_accd<2>(cov, T_, D, K, X, W, acc, part, cpart, s);
        case 3: return launch_accd<3>(cov, T_, D, K, X, W, acc, part, cpart, s);
        default: return launch_accd<4>(cov, T_, D, K, X, W, acc, part, cpart, s);
    }
}

}  // namespace beer_mfma
