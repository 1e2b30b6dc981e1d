// Shared pieces of the matrix-core E-step kernels (estep_mfma.hip: exact
// fp32 / fp64 MFMA; estep_f16.hip: fp32 operands split into two fp16 halves):
// MFMA traits, DPP row reductions, the slab enumeration of the contraction
// index and the in-register softmax epilogue.  See estep_mfma.hip for the design.
#pragma once

#include <hip/hip_runtime.h>

#include "common.h"

namespace beer_mfma {

using namespace beer;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <typename T> struct Mma;
template <> struct Mma<float> {
    using acc_t = f32x4;
    using vec4_t = f32x4;
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int g, int r) { return 4 * g + r; }
    // exp of a non-positive softmax argument: v_exp_f32 (1 ulp) on x * log2(e).  The
    // product's rounding adds |x| * 6e-8 of relative error to a term of weight
    // e^x -- at most 2e-8 of the sum -- against the 12 instructions of expf.
    static __device__ __forceinline__ float exp_neg(float x) {
        return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f);
    }
    static __device__ __forceinline__ float recip(float x) { return __builtin_amdgcn_rcpf(x); }
    // v_max_f32 (one instruction, also as a DPP operand) instead of compare + select
    static __device__ __forceinline__ float mx(float a, float b) { return __builtin_fmaxf(a, b); }
    // log of a softmax denominator (a sum in [1, group size]): v_log_f32 (log2, 1 ulp)
    static __device__ __forceinline__ float log_sum(float x) {
        return __builtin_amdgcn_logf(x) * 0.69314718055994530942f;
    }
};
template <> struct Mma<double> {
    using acc_t = f64x4;
    using vec4_t = f64x4;
    static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int g, int r) { return g + 4 * r; }
    static __device__ __forceinline__ double exp_neg(double x) { return exp(x); }
    static __device__ __forceinline__ double recip(double x) { return 1.0 / x; }
    static __device__ __forceinline__ double mx(double a, double b) { return b > a ? b : a; }
    static __device__ __forceinline__ double log_sum(double x) { return log(x); }
};

// All-reduce across the 16 lanes that hold one row of a 16x16 C tile, with
// DPP (VALU, no LDS round trip): xor-1 / xor-2 inside a quad, then the two
// mirrors (values are uniform inside a quad / half-row by then, so mirroring
// equals the xor-4 / xor-8 exchange).  `gl` = lanes per group (1,2,4,8,16).
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
        0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
template <typename T>
__device__ __forceinline__ T group_max(T v, int gl) {
    if (gl > 1) v = Mma<T>::mx(v, dpp_move<0xB1>(v));      // quad_perm [1,0,3,2]
    if (gl > 2) v = Mma<T>::mx(v, dpp_move<0x4E>(v));      // quad_perm [2,3,0,1]
    if (gl > 4) v = Mma<T>::mx(v, dpp_move<0x141>(v));     // row_half_mirror
    if (gl > 8) v = Mma<T>::mx(v, dpp_move<0x140>(v));     // row_mirror
    return v;
}
template <typename T>
__device__ __forceinline__ T group_sum(T v, int gl) {
    if (gl > 1) v += dpp_move<0xB1>(v);
    if (gl > 2) v += dpp_move<0x4E>(v);
    if (gl > 4) v += dpp_move<0x141>(v);
    if (gl > 8) v += dpp_move<0x140>(v);
    return v;
}

__host__ __device__ inline int d4_of(int D) { return (D + 3) / 4; }

// Slab enumeration per covariance type (see the header comment).  Diagonal and
// isotropic models only have D4 "square" slabs xe[4j+g]^2 (flag bit 16 in the
// table), D4 linear slabs and the constant slab.
__host__ __device__ inline int nslab_of(int cov, int D) {
    const int D4 = d4_of(D);
    int n = D4 + 1;                                   // linear + constant
    if (cov != BEER_FULL) return n + D4;
    for (int a = 0; a < D; ++a) n += D4 - a / 4;
    return n;
}
// slab s -> table entry  a | (4j << 8) | (square << 16)
__host__ __device__ inline int slab_entry(int cov, int D, int s) {
    const int D4 = d4_of(D), Dp = 4 * D4, nslab = nslab_of(cov, D);
    const int nquad = nslab - (D4 + 1);
    if (s >= nslab) return (Dp + 1) | (Dp << 8);          // padding: zero column
    if (s >= nquad) return Dp | ((4 * (s - nquad)) << 8);  // linear / constant
    if (cov != BEER_FULL) return Dp | ((4 * s) << 8) | (1 << 16);
    int rem = s, a = 0;
    for (;;) { const int len = D4 - a / 4; if (rem < len) break; rem -= len; ++a; }
    return a | ((4 * (a / 4 + rem)) << 8);
}
// index of the quadratic slab holding x_a * x_b (a <= b), of linear slab j
// (a == Dp) and of the constant slab (a == Dp, j == D4)
__host__ __device__ inline int slab_index(int cov, int D, int a, int j) {
    const int D4 = d4_of(D);
    if (a >= D) return nslab_of(cov, D) - (D4 + 1) + j;
    if (cov != BEER_FULL) return j;                       // square slab of x_{4j..4j+3}
    const int q = a / 4, r = a % 4;
    const int before = 4 * (q * D4 - q * (q - 1) / 2) + r * (D4 - q);
    return before + (j - q);
}

// slabs in the packed parameter image: an even count (the K1 loop is unrolled
// by two) plus one look-ahead slab, all zero beyond nslab_of().
__host__ __device__ inline int nslab_padded(int cov, int D) {
    return (nslab_of(cov, D) + 1) / 2 * 2 + 1;
}

constexpr int kThreads = 256;
constexpr double kPadLogit = -1.0e30;

namespace {

__global__ void tab_kernel(int cov, int D, int* __restrict__ tab) {
    const int nslab = nslab_of(cov, D);
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < nslab; s += gridDim.x * blockDim.x)
        tab[s] = slab_entry(cov, D, s);
}

// Sp [K][nslab*4] (packed sums) -> acc [K][Q] += in the reference's layout:
// full [sum r x, -.5 sum r x x^T (dense D x D), -.5 N, +.5 N], diagonal
// [sum r x, -.5 sum r x^2, -.5 N, +.5 N], isotropic [sum r x, -.5 sum r |x|^2,
// -.5 N, +.5 D N].
__global__ void unpack_kernel(int cov, int D, int K, const double* __restrict__ Sp,
                              double* __restrict__ acc) {
    const int D4 = d4_of(D), Dp = 4 * D4, nq = nslab_of(cov, D) * 4;
    const int Q = stats_dim(cov, D);
    const int64_t total = (int64_t)K * Q;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(idx / Q), q = (int)(idx % Q);
        const double* s = Sp + (size_t)k * nq;
        double v;
        if (q < D) {
            v = s[slab_index(cov, D, Dp, q / 4) * 4 + q % 4];
        } else if (q >= Q - 2) {
            const double n = s[slab_index(cov, D, Dp, D4) * 4];
            v = (q == Q - 2) ? -0.5 * n : (cov == BEER_ISO ? 0.5 * (double)D * n : 0.5 * n);
        } else if (cov == BEER_FULL) {
            int a = (q - D) / D, b = (q - D) % D;
            if (a > b) { const int t = a; a = b; b = t; }
            v = -0.5 * s[slab_index(cov, D, a, b / 4) * 4 + b % 4];
        } else if (cov == BEER_DIAG) {
            const int d = q - D;
            v = -0.5 * s[slab_index(cov, D, d, d / 4) * 4 + d % 4];
        } else {
            double tot = 0.0;
            for (int d = 0; d < D; ++d) tot += s[slab_index(cov, D, d, d / 4) * 4 + d % 4];
            v = -0.5 * tot;
        }
        acc[idx] += v;
    }
}

}  // namespace

// ---------------------------------------------------------------------------
// Epilogue of K1: logsumexp + responsibilities over each group of G components
// of a wave's acc[MT][NT] tiles (frames fb .. fb + 16 MT, components kbase ..).
// ---------------------------------------------------------------------------
// PACKED (float only): instead of float32 responsibilities, `resps` receives
// them as the accumulation kernel's LDS image, so that it copies them there
// without arithmetic.  The image is cut into tiles of 64 frames x 128
// components (tile index tau * nblk + beta, 32 KB each): first the fp16 high
// halves of r * 2^12 as rows [component][64 frames] of 128 bytes whose 16-byte
// chunks (8 frames) are stored at position chunk ^ (component & 7) -- the
// swizzle that makes the MFMA fragment reads conflict-free -- then the low
// halves in the same arrangement.  Frames past T and components past K are 0.
constexpr int kPackedRespBits = 12;
constexpr int kPackedFrames = 64, kPackedComps = 128;

// 32-bit word index of the hi half of (component kk of block beta, frame f6 of
// tile tau); the lo half is kPackedComps * kPackedFrames / 2 words further
__host__ __device__ inline size_t packed_word(int64_t tau, int nblk, int beta, int kk, int f6) {
    const int half = kk * kPackedFrames + (((f6 >> 3) ^ (kk & 7)) << 3) + (f6 & 7);
    return ((size_t)tau * nblk + beta) * (kPackedComps * kPackedFrames) + (half >> 1);
}

// LNO: only the log-normalisers are wanted (the accumulation recomputes the
// responsibilities): no reciprocal, no normalisation, no store of them.
template <typename T, int NT, int MT, int GQ, bool PACKED = false, bool LNO = false>
__device__ __forceinline__ void softmax_epilogue(
    typename Mma<T>::acc_t (&acc)[MT][NT], int64_t fb, int64_t nframes, int kbase, int K, int S,
    int G, int gl, int jw, int i, int g, int lane, T* __restrict__ resps,
    T* __restrict__ log_norm, double* __restrict__ llh_sum) {
    using M = Mma<T>;
    using vec4_t = typename M::vec4_t;
    double llh_local = 0.0;
    const bool vec_ok = (K % 4) == 0;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // one row at a time: without the barrier hipcc interleaves all the
            // rows and groups and spills the accumulators to scratch
            __builtin_amdgcn_sched_barrier(0);
            const int64_t f = fb + m * 16 + M::row(g, r);
#pragma unroll
            for (int tq = 0; tq < NT / 4 / GQ; ++tq) {
                T e[GQ][4];
                const int kq = kbase + 64 * tq * GQ + 4 * i;       // lane's first component
                if (jw == 4) {
                    T mx = acc[m][4 * tq * GQ][r];
#pragma unroll
                    for (int c = 1; c < 4 * GQ; ++c) mx = M::mx(mx, acc[m][4 * tq * GQ + c][r]);
                    mx = group_max(mx, gl);
                    T sum = 0;
#pragma unroll
                    for (int qq = 0; qq < GQ; ++qq)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            e[qq][j] = M::exp_neg(acc[m][4 * (tq * GQ + qq) + j][r] - mx);
                            sum += e[qq][j];
                        }
                    sum = group_sum(sum, gl);
                    const T lse = mx + M::log_sum(sum);
                    if (!LNO) {
                        const T inv = M::recip(sum);
#pragma unroll
                        for (int qq = 0; qq < GQ; ++qq)
#pragma unroll
                            for (int j = 0; j < 4; ++j) e[qq][j] *= inv;
                    }
                    const int state = (kbase + 64 * tq * GQ + 4 * (i & ~(gl - 1))) / G;
                    if (f < nframes && state < S && (i & (gl - 1)) == 0) {
                        if (log_norm) log_norm[f * S + state] = lse;
                        llh_local += (double)lse;
                    }
                } else {
                    // G = 1 or 2: groups inside a lane's 4 values (GQ == 1)
#pragma unroll
                    for (int j0 = 0; j0 < 4; j0 += 2) {
                        const T a0 = acc[m][4 * tq * GQ + j0][r], a1 = acc[m][4 * tq * GQ + j0 + 1][r];
                        if (jw == 2) {
                            const T mx = a0 > a1 ? a0 : a1;
                            const T e0 = M::exp_neg(a0 - mx), e1 = M::exp_neg(a1 - mx);
                            const T lse = mx + M::log_sum(e0 + e1);
                            e[0][j0] = e0 / (e0 + e1);
                            e[0][j0 + 1] = e1 / (e0 + e1);
                            const int state = (kq + j0) / G;
                            if (f < nframes && state < S) {
                                if (log_norm) log_norm[f * S + state] = lse;
                                llh_local += (double)lse;
                            }
                        } else {
                            e[0][j0] = 1;
                            e[0][j0 + 1] = 1;
                            if (f < nframes) {
                                if (kq + j0 < K) {
                                    if (log_norm) log_norm[f * S + kq + j0] = a0;
                                    llh_local += (double)a0;
                                }
                                if (kq + j0 + 1 < K) {
                                    if (log_norm) log_norm[f * S + kq + j0 + 1] = a1;
                                    llh_local += (double)a1;
                                }
                            }
                        }
                    }
                }
                if (PACKED) {
                    // keep the responsibilities where the logits were; packed below
#pragma unroll
                    for (int qq = 0; qq < GQ; ++qq)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[m][4 * (tq * GQ + qq) + j][r] = e[qq][j];
                } else if (!LNO && resps && f < nframes) {
#pragma unroll
                    for (int qq = 0; qq < GQ; ++qq) {
                        const int k = kq + 64 * qq;
                        T* dst = resps + f * K + k;
                        if (vec_ok && k + 3 < K) {
                            // (streamed: written once, read once by the accumulation)
                            __builtin_nontemporal_store(
                                vec4_t{e[qq][0], e[qq][1], e[qq][2], e[qq][3]},
                                reinterpret_cast<vec4_t*>(dst));
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (k + j < K) dst[j] = e[qq][j];
                        }
                    }
                }
            }
        }
    }
    if constexpr (PACKED) {
        // acc[m][4 tt + j] = component kbase + 64 tt + 4 i + j, its 4 values = frames
        // fb + 16 m + 4 g + (0..3): half of a 16-byte chunk (8 frames) of the image
        // row.  Lanes g and g ^ 1 (16 lanes apart) hold the two halves, for m = 0
        // and for m = 1: one v_permlane16_swap per word leaves the even-g lane with
        // the whole chunk of m = 0 and the odd-g lane with that of m = 1, so that
        // every lane stores 16 bytes and 4 lanes fill a 64-byte segment of the row.
        static_assert(!PACKED || MT % 2 == 0, "the lane pairs exchange two frame tiles");
        typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
        typedef float float4_t __attribute__((ext_vector_type(4)));
        typedef unsigned int uint2_t __attribute__((ext_vector_type(2)));
        typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));
        unsigned int* out = reinterpret_cast<unsigned int*>(resps);
        const float up = (float)(1 << kPackedRespBits);
        const int nblk = (K + kPackedComps - 1) / kPackedComps;
        const int64_t tiles = (nframes + kPackedFrames - 1) / kPackedFrames;
#pragma unroll
        for (int mp = 0; mp < MT / 2; ++mp) {
            // the chunk this lane stores after the exchange
            const int64_t fc = fb + 32 * mp + 16 * (g & 1) + 8 * (g >> 1);
            const int64_t tau = fc / kPackedFrames;
            const int f6 = (int)(fc - tau * kPackedFrames);
            const bool chunk_ok = out && fc < tiles * kPackedFrames;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                __builtin_amdgcn_sched_barrier(0);
                const int k = kbase + 64 * (nt >> 2) + 4 * i + (nt & 3);
                uint2_t hi[2], lo[2];
#pragma unroll
                for (int mm = 0; mm < 2; ++mm) {
                    const int m = 2 * mp + mm;
                    const int64_t f0 = fb + m * 16 + M::row(g, 0);
                    float4_t v;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        v[r] = f0 + r < nframes && k < K ? (float)acc[m][nt][r] * up : 0.f;
                    const half4_t h = __builtin_convertvector(v, half4_t);
                    const half4_t l = __builtin_convertvector(
                        v - __builtin_convertvector(h, float4_t), half4_t);
                    hi[mm] = __builtin_bit_cast(uint2_t, h);
                    lo[mm] = __builtin_bit_cast(uint2_t, l);
                }
                uint4_t ch, cl;
#pragma unroll
                for (int wd = 0; wd < 2; ++wd) {
                    const auto sh = __builtin_amdgcn_permlane16_swap(hi[0][wd], hi[1][wd], false, false);
                    const auto sl = __builtin_amdgcn_permlane16_swap(lo[0][wd], lo[1][wd], false, false);
                    ch[wd] = sh[0]; ch[2 + wd] = sh[1];
                    cl[wd] = sl[0]; cl[2 + wd] = sl[1];
                }
                if (chunk_ok && k < nblk * kPackedComps) {
                    unsigned int* dst = out + packed_word(tau, nblk, k / kPackedComps,
                                                          k & (kPackedComps - 1), f6);
                    // non-temporal: streamed once; keeps the packed parameters in L2
                    __builtin_nontemporal_store(ch, reinterpret_cast<uint4_t*>(dst));
                    __builtin_nontemporal_store(
                        cl, reinterpret_cast<uint4_t*>(dst + kPackedComps * kPackedFrames / 2));
                }
            }
        }
    }
    if (llh_sum) {
        llh_local = wave_sum(llh_local);
        if (lane == 0) atomicAdd(llh_sum, llh_local);
    }
}

// ---------------------------------------------------------------------------
// Epilogue of K1 when TWO waves share the frames of a tile and own one half of
// the components each (one mixture, float, packed output): the logsumexp over
// all components needs the partner's partial maximum and partial sum, exchanged
// through LDS (`xch`: 2 x 4 waves x 16 MT floats; two workgroup barriers).
// acc[m][nt] of a wave = component kbase + 64 (nt / 4) + 4 i + nt % 4 (kbase
// includes the wave's half), frames fb + 16 m + 4 g + (0..3).  Writes the packed
// tiles (see PACKED above), log_norm and llh_sum (from the even wave).
// ---------------------------------------------------------------------------
template <int NT, int MT>
__device__ __forceinline__ void softmax_epilogue_pair(
    f32x4 (&acc)[MT][NT], int64_t fb, int64_t nframes, int kbase, int K, int i, int g, int lane,
    int wave, float* __restrict__ xch, float* __restrict__ resps, float* __restrict__ log_norm,
    double* __restrict__ llh_sum) {
    using M = Mma<float>;
    static_assert(MT % 2 == 0, "lane pairs exchange two frame tiles");
    constexpr int ROWS = 16 * MT;
    float* mine = xch + wave * ROWS;
    const float* theirs = xch + (wave ^ 1) * ROWS;
    float mx[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = acc[m][0][r];
#pragma unroll
            for (int nt = 1; nt < NT; ++nt) v = acc[m][nt][r] > v ? acc[m][nt][r] : v;
            v = group_max(v, 16);
            mx[m][r] = v;
            if (i == 0) mine[m * 16 + M::row(g, r)] = v;
        }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float o = theirs[m * 16 + M::row(g, r)];
            mx[m][r] = o > mx[m][r] ? o : mx[m][r];
        }
    float* mine_s = mine + 4 * ROWS;
    const float* theirs_s = theirs + 4 * ROWS;
    float sum[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            __builtin_amdgcn_sched_barrier(0);
            float sacc = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float e = M::exp_neg(acc[m][nt][r] - mx[m][r]);
                acc[m][nt][r] = e;
                sacc += e;
            }
            sacc = group_sum(sacc, 16);
            sum[m][r] = sacc;
            if (i == 0) mine_s[m * 16 + M::row(g, r)] = sacc;
        }
    __syncthreads();
    double llh_local = 0.0;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float total = sum[m][r] + theirs_s[m * 16 + M::row(g, r)];
            const float inv = M::recip(total);
            const int64_t f = fb + m * 16 + M::row(g, r);
            if ((wave & 1) == 0 && i == 0 && f < nframes) {
                const float lse = mx[m][r] + M::log_sum(total);
                if (log_norm) log_norm[f] = lse;
                llh_local += (double)lse;
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[m][nt][r] *= inv;
        }
    // packing: as in softmax_epilogue<PACKED>, per pair of frame tiles
    typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
    typedef float float4_t __attribute__((ext_vector_type(4)));
    typedef unsigned int uint2_t __attribute__((ext_vector_type(2)));
    typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));
    unsigned int* out = reinterpret_cast<unsigned int*>(resps);
    const float up = (float)(1 << kPackedRespBits);
    const int nblk = (K + kPackedComps - 1) / kPackedComps;
    const int64_t tiles = (nframes + kPackedFrames - 1) / kPackedFrames;
#pragma unroll
    for (int mp = 0; mp < MT / 2; ++mp) {
        const int64_t fc = fb + 32 * mp + 16 * (g & 1) + 8 * (g >> 1);
        const int64_t tau = fc / kPackedFrames;
        const int f6 = (int)(fc - tau * kPackedFrames);
        const bool chunk_ok = out && fc < tiles * kPackedFrames;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            __builtin_amdgcn_sched_barrier(0);
            const int k = kbase + 64 * (nt >> 2) + 4 * i + (nt & 3);
            uint2_t hi[2], lo[2];
#pragma unroll
            for (int mm = 0; mm < 2; ++mm) {
                const int m = 2 * mp + mm;
                const int64_t f0 = fb + m * 16 + M::row(g, 0);
                float4_t v;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    v[r] = f0 + r < nframes && k < K ? acc[m][nt][r] * up : 0.f;
                const half4_t h = __builtin_convertvector(v, half4_t);
                const half4_t l = __builtin_convertvector(
                    v - __builtin_convertvector(h, float4_t), half4_t);
                hi[mm] = __builtin_bit_cast(uint2_t, h);
                lo[mm] = __builtin_bit_cast(uint2_t, l);
            }
            uint4_t ch, cl;
#pragma unroll
            for (int wd = 0; wd < 2; ++wd) {
                const auto sh = __builtin_amdgcn_permlane16_swap(hi[0][wd], hi[1][wd], false, false);
                const auto sl = __builtin_amdgcn_permlane16_swap(lo[0][wd], lo[1][wd], false, false);
                ch[wd] = sh[0]; ch[2 + wd] = sh[1];
                cl[wd] = sl[0]; cl[2 + wd] = sl[1];
            }
            if (chunk_ok && k < nblk * kPackedComps) {
                unsigned int* dst = out + packed_word(tau, nblk, k / kPackedComps,
                                                      k & (kPackedComps - 1), f6);
                __builtin_nontemporal_store(ch, reinterpret_cast<uint4_t*>(dst));
                __builtin_nontemporal_store(
                    cl, reinterpret_cast<uint4_t*>(dst + kPackedComps * kPackedFrames / 2));
            }
        }
    }
    if (llh_sum) {
        llh_local = wave_sum(llh_local);
        if (lane == 0 && (wave & 1) == 0) atomicAdd(llh_sum, llh_local);
    }
}

}  // namespace beer_mfma
