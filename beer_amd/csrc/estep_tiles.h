// Shared pieces of the matrix-core E-step kernels (estep_mfma.hip: exact
// fp32 / fp64 MFMA; estep_bf16.hip: fp32 operands as three bf16 pieces):
// MFMA traits, DPP row reductions, the slab enumeration of the contraction
// index and the in-register softmax epilogue.  See estep_mfma.hip for the design.
#pragma once
#include <type_traits>

#include <hip/hip_runtime.h>

#include "common.h"

namespace beer_mfma {

using namespace beer;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

// 2^y, y <~ 0, on the plain vector ALU.  v_exp_f32 runs at a quarter of the vector rate AND
// keeps the matrix pipe of its SIMD from issuing meanwhile -- measured
// (tools/probes/coissue.hip): a wave of fma + add + exp triples beside a wave of MFMAs on the
// same SIMD takes the SUM of their times (65.6 k cycles against 35.4 k + 35 k), where plain
// fmas vanish behind the MFMAs (37 k); in the epilogues below that was 16 cycles of matrix
// pipe per exponential.  y = n + f, n = rint(y) by the 1.5 * 2^23 trick, 2^f on [-1/2, 1/2] by a
// degree-6 polynomial (max relative error 7.9e-8 evaluated in float32: a float32 ulp, like
// v_exp_f32), 2^n added into the exponent field.  Arguments below -125 (padded components,
// rows past the end) give 2^-125; NaN arguments give 2^-125 too (the log-normaliser of the
// frame is NaN regardless: its maximum is).  11 full-rate instructions.
#ifndef BEER_EXP_VALU
#define BEER_EXP_VALU 0
#endif
#ifndef BEER_LNFI_TRANS_EVERY
#define BEER_LNFI_TRANS_EVERY 0     // lane-major log-normaliser epilogue: every n-th exponential on v_exp_f32 (0: none)
#endif
__device__ __forceinline__ float exp2_valu(float y) {
    if (!BEER_EXP_VALU) return __builtin_amdgcn_exp2f(y);
    y = __builtin_fmaxf(y, -125.f);
    const float magic = 12582912.f;
    const float t = y + magic;
    const float f = y - (t - magic);
    float p = 0.00015345810970757157f;
    p = __builtin_fmaf(p, f, 0.0013399930903688073f);
    p = __builtin_fmaf(p, f, 0.009618489071726799f);
    p = __builtin_fmaf(p, f, 0.05550328642129898f);
    p = __builtin_fmaf(p, f, 0.24022646248340607f);
    p = __builtin_fmaf(p, f, 0.6931471824645996f);
    p = __builtin_fmaf(p, f, 1.0f);
    return __builtin_bit_cast(float, __builtin_bit_cast(int, p) +
                                         (int)((unsigned)__builtin_bit_cast(int, t) << 23));
}

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

// Slab enumeration per covariance type (see the header comment of estep_mfma.hip).
//
// Diagonal and isotropic models have D4 "square" slabs xe[4q+g]^2 (flag bit 16 in the
// table), D4 linear slabs 1 * xe[4q+g] and constant slabs 1 * (1, eps, 0, 0).  Their
// order is chosen for the ACCUMULATOR: the terms of a logit, -1/2 lambda x^2 +
// lambda mu x - 1/2 lambda mu^2, are each an order of magnitude larger than their
// sum, and every accumulation rounds (the bf16 MFMA truncates) at the size of the
// running sum.  So the square and the linear slab of the same four dimensions are
// neighbours, and every group of 8 slabs (one k-step of the 32-deep MFMA) is closed
// by a constant slab that carries the -1/2 lambda mu^2 of the dimensions seen so far
// (`const_share`): after every k-step the accumulator holds
// sum_d -1/2 lambda_d (x_d - mu_d)^2 -- small for the components that matter.
//   kind 0 = square, 1 = linear, 2 = closing constant, 3 = final constant
__host__ __device__ inline int diag_walk(int D, int s_query, int* kind, int* quad) {
    // slab s_query -> (kind, quad); returns the slab count when s_query is not a slab
    const int nitems = 2 * d4_of(D);
    int it = 0;
    for (int s = 0;; ++s) {
        int k, q = 0;
        if (it == nitems) k = 3;
        else if ((s & 7) == 7) k = 2;
        else { k = it & 1; q = it >> 1; ++it; }
        if (s == s_query) { *kind = k; *quad = q; return s; }
        if (k == 3) return s + 1;
    }
}
__host__ __device__ inline int nslab_of(int cov, int D) {
    const int D4 = d4_of(D);
    if (cov != BEER_FULL) { int k, q; return diag_walk(D, -1, &k, &q); }
    int n = D4 + 1;                                   // linear + constant
    for (int a = 0; a < D; ++a) n += D4 - a / 4;
    return n;
}
// slab s -> table entry  a | (4j << 8) | (square << 16)
__host__ __device__ inline int slab_entry(int cov, int D, int s) {
    const int D4 = d4_of(D), Dp = 4 * D4, nslab = nslab_of(cov, D);
    if (s >= nslab) return (Dp + 2) | (Dp << 8);          // padding: zero column
    if (cov != BEER_FULL) {
        int k, q;
        diag_walk(D, s, &k, &q);
        if (k == 0) return Dp | ((4 * q) << 8) | (1 << 16);
        return k == 1 ? (Dp | ((4 * q) << 8)) : (Dp | (Dp << 8));
    }
    const int nquad = nslab - (D4 + 1);
    if (s >= nquad) return Dp | ((4 * (s - nquad)) << 8);  // linear / constant
    int rem = s, a = 0;
    for (;;) { const int len = D4 - a / 4; if (rem < len) break; rem -= len; ++a; }
    return a | ((4 * (a / 4 + rem)) << 8);
}
// index of the quadratic slab holding x_a * x_b (a <= b), of linear slab j
// (a == Dp) and of the (final) constant slab (a == Dp, j == D4)
__host__ __device__ inline int slab_index(int cov, int D, int a, int j) {
    const int D4 = d4_of(D);
    if (cov != BEER_FULL) {
        const int nslab = nslab_of(cov, D);
        if (a >= D && j >= D4) return nslab - 1;
        const int want = a >= D ? 1 : 0;                 // linear / square slab of quad j
        for (int s = 0; s < nslab; ++s) {
            int k, q;
            diag_walk(D, s, &k, &q);
            if (k == want && q == j) return s;
        }
        return nslab - 1;
    }
    if (a >= D) return nslab_of(cov, D) - (D4 + 1) + j;
    const int q = a / 4, r = a % 4;
    const int before = 4 * (q * D4 - q * (q - 1) / 2) + r * (D4 - q);
    return before + (j - q);
}
// The constant term of component `row` (E[T], any float type) that constant slab
// `slab` carries.  Full covariance: one slab, the whole constant.  Diagonal /
// isotropic: a closing slab holds -1/2 sum_d (lambda mu)_d^2 / lambda_d over the
// dimensions whose linear slab lies between the previous constant slab and itself;
// the final one the rest, so that the shares add up to the constant exactly.
template <typename T>
__host__ __device__ inline double const_total(int cov, int D, const T* row, double logw) {
    const int Q = stats_dim(cov, D);
    const double zero = cov == BEER_ISO ? 0.5 * (double)D : 0.5;
    return -0.5 * (double)row[Q - 2] + zero * (double)row[Q - 1] - 0.5 * (double)D * kLog2Pi + logw;
}
template <typename T>
__host__ __device__ inline double const_share(int cov, int D, int slab, const T* row, double logw) {
    const double total = const_total(cov, D, row, logw);
    if (cov == BEER_FULL) return total;
    double given = 0.0, pending = 0.0;
    const int nslab = nslab_of(cov, D);
    for (int s = 0; s < nslab; ++s) {
        int k, q;
        diag_walk(D, s, &k, &q);
        if (k == 1) {
            for (int d = 4 * q; d < 4 * q + 4 && d < D; ++d) {
                const double lam = (double)row[cov == BEER_ISO ? D : D + d], lm = (double)row[d];
                if (lam > 0.0) pending -= 0.5 * lm * lm / lam;
            }
        } else if (k >= 2) {
            if (s == slab) return k == 3 ? total - given : pending;
            given += pending;
            pending = 0.0;
        }
    }
    return 0.0;
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
// without arithmetic.  Every responsibility is held EXACTLY as three bf16 pieces,
// r = p0 + p1 + p2 (split3 below: 3 x 8 significand bits = the 24 of float32, bf16
// has float32's exponent range, so there is no scaling and nothing to undo).  The
// image is cut into tiles of 64 frames x 128 components (tile index
// tau * nblk + beta, 48 KB each): three planes of 16 KB, plane q = the pieces p_q
// as rows [component][64 frames] of 128 bytes whose 16-byte chunks (8 frames) are
// stored at position chunk ^ (component & 7) -- the swizzle that makes the MFMA
// fragment reads conflict-free.  Frames past T and components past K are 0.
constexpr int kPackedFrames = 64, kPackedComps = 128, kPackedPieces = 3;
constexpr int kPackedPlaneWords = kPackedComps * kPackedFrames / 2;       // 32-bit words

// 32-bit word index of piece 0 of (component kk of block beta, frame f6 of tile
// tau); piece q is q * kPackedPlaneWords words further
__host__ __device__ inline size_t packed_word(int64_t tau, int nblk, int beta, int kk, int f6) {
    const int half = kk * kPackedFrames + (((f6 >> 3) ^ (kk & 7)) << 3) + (f6 & 7);
    return ((size_t)tau * nblk + beta) * (kPackedPieces * kPackedPlaneWords) + (half >> 1);
}

// A pair of float32 values as three words of two bf16 each, v = p0 + p1 + p2
// EXACTLY: p0 = bf16(v) (round to nearest even, v_cvt_pk_bf16_f32), the remainder
// v - p0 is exact in float32 and has at most 16 significant bits, p1 = bf16 of it,
// the second remainder has at most 8 and is a bf16 itself.  Rounding to nearest
// (not truncation) keeps the pieces' signs independent, so that the products a
// six-term multiplication drops (p1 q2 + p2 p1', 2^-24 of the product) carry no
// systematic sign.  The low half of a word is the first value of the pair.
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
    typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
    typedef float f2_t __attribute__((ext_vector_type(2)));
    const f2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2_t));
}
__device__ __forceinline__ void split3(float a, float b, unsigned (&p)[3]) {
    // (no contraction: `a` is usually a product, and a - p0 fused with it into an fma would
    // decompose the UNROUNDED product -- harmless, but then two kernels that hipcc
    // happens to compile differently no longer produce the same bits, and the fused
    // accumulation relies on recomputing the E-step's logits exactly.  The pieces
    // decompose fl32(x_a x_b), the reference's own statistic.)
#pragma clang fp contract(off)
    p[0] = cvt_pk_bf16(a, b);
    float ra = a - __builtin_bit_cast(float, p[0] << 16);
    float rb = b - __builtin_bit_cast(float, p[0] & 0xffff0000u);
    p[1] = cvt_pk_bf16(ra, rb);
    ra -= __builtin_bit_cast(float, p[1] << 16);
    rb -= __builtin_bit_cast(float, p[1] & 0xffff0000u);
    p[2] = cvt_pk_bf16(ra, rb);
}
// split3 one instruction group at a time (k = 0 .. 6), for kernels that place the
// fragment arithmetic by hand between their MFMAs
// (`pin`: an empty volatile asm that redefines the value.  Volatile asms keep their
// order, so a value pinned between two pinned MFMAs is computed between them --
// hipcc's IR passes otherwise sink the whole computation to its first use, a k-step
// later, and keep its inputs alive until then.)
template <typename V>
__device__ __forceinline__ void pin(V& v) { asm volatile("" : "+v"(v)); }
struct Split3Steps { unsigned w0, w1, w2; float r0, r1; };
// BEER_ASM_STEPS: every step is a volatile asm statement (one or two instructions).  Pinning
// the RESULT of a C++ step (pin()) keeps it from sinking to its first use, but the machine
// scheduler still slides the instruction itself along the stream and gathers the steps into
// runs of four or five behind every second MFMA -- and an in-order wave that meets
// `M M v v v v` waits for the matrix pipe in front of the second M with nothing to issue
// (21.5 cycles per MFMA against 17.3 for `M v v M v v`, tools/probes/coissue.hip).  Volatile asm
// statements keep their program order among themselves: the steps stay where they are written.
#ifndef BEER_ASM_STEPS
#define BEER_ASM_STEPS 1
#endif
__device__ __forceinline__ void split3_step(int k, float a, float b, Split3Steps& t) {
#pragma clang fp contract(off)
    if (BEER_ASM_STEPS) {
        unsigned tmp;
        switch (k) {
            case 0: asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(t.w0) : "v"(a), "v"(b)); break;
            case 1: asm volatile("v_lshlrev_b32 %0, 16, %1\n\tv_sub_f32 %0, %2, %0"
                                 : "=&v"(t.r0) : "v"(t.w0), "v"(a)); break;
            case 2: asm volatile("v_and_b32 %0, 0xffff0000, %1\n\tv_sub_f32 %0, %2, %0"
                                 : "=&v"(t.r1) : "v"(t.w0), "v"(b)); break;
            case 3: asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(t.w1) : "v"(t.r0), "v"(t.r1)); break;
            case 4: asm volatile("v_lshlrev_b32 %0, 16, %2\n\tv_sub_f32 %1, %1, %0"
                                 : "=&v"(tmp), "+v"(t.r0) : "v"(t.w1)); break;
            case 5: asm volatile("v_and_b32 %0, 0xffff0000, %2\n\tv_sub_f32 %1, %1, %0"
                                 : "=&v"(tmp), "+v"(t.r1) : "v"(t.w1)); break;
            case 6: asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(t.w2) : "v"(t.r0), "v"(t.r1)); break;
            default: break;
        }
        return;
    }
    switch (k) {
        case 0: t.w0 = cvt_pk_bf16(a, b); pin(t.w0); break;
        case 1: t.r0 = a - __builtin_bit_cast(float, t.w0 << 16); pin(t.r0); break;
        case 2: t.r1 = b - __builtin_bit_cast(float, t.w0 & 0xffff0000u); pin(t.r1); break;
        case 3: t.w1 = cvt_pk_bf16(t.r0, t.r1); pin(t.w1); break;
        case 4: t.r0 -= __builtin_bit_cast(float, t.w1 << 16); pin(t.r0); break;
        case 5: t.r1 -= __builtin_bit_cast(float, t.w1 & 0xffff0000u); pin(t.r1); break;
        case 6: t.w2 = cvt_pk_bf16(t.r0, t.r1); pin(t.w2); break;
        default: break;
    }
}
// the same on the host / in scalar device code (parameter packing): bf16 bits
__host__ __device__ inline unsigned short bf16_rne(float v) {
    unsigned u = __builtin_bit_cast(unsigned, v);
    if ((u & 0x7f800000u) == 0x7f800000u) return (unsigned short)(u >> 16);   // inf / nan
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__host__ __device__ inline void split3_scalar(float v, unsigned short (&p)[3]) {
    p[0] = bf16_rne(v);
    float r = v - __builtin_bit_cast(float, (unsigned)p[0] << 16);
    if (!(r == r) || r - r != 0.f) r = 0.f;                 // v = +-inf: the first piece holds it
    p[1] = bf16_rne(r);
    r -= __builtin_bit_cast(float, (unsigned)p[1] << 16);
    p[2] = bf16_rne(r);
}

// LNO: only the log-normalisers are wanted (the accumulation recomputes the
// responsibilities): no reciprocal, no normalisation, no store of them.
template <typename T, int NT, int MT, int GQ, bool PACKED = false, bool LNO = false>
__device__ __forceinline__ void softmax_epilogue(
    typename Mma<T>::acc_t (&acc)[MT][NT], int64_t fb, int64_t nframes, int kbase, int K, int S,
    int G, int gl, int jw, int i, int g, int lane, T* __restrict__ resps,
    T* __restrict__ log_norm, double* __restrict__ llh_sum, T shift = 0) {
    // `shift`: a constant the caller took out of every logit (softmax is invariant to
    // it); it goes back into the log-normalisers here
    using M = Mma<T>;
    using vec4_t = typename M::vec4_t;
    double llh_local = 0.0;
    const bool vec_ok = (K % 4) == 0;
    // (the state of the lane's group per group of tiles, and whether this lane writes its
    // log-normaliser: neither depends on the row -- an integer division by a run-time G each, which
    // the scheduling barriers below kept inside the row loop, eight times per tile)
    int state_of[NT / 4 / GQ];
    bool writer[NT / 4 / GQ];
#pragma unroll
    for (int tq = 0; tq < NT / 4 / GQ; ++tq) {
        state_of[tq] = jw == 4 ? (kbase + 64 * tq * GQ + 4 * (i & ~(gl - 1))) / G : 0;
        writer[tq] = state_of[tq] < S && (i & (gl - 1)) == 0;
    }
    // (frames fb .. fb + 16 MT - 1 all inside the launch: no row needs its index compared)
    const bool rows_in = fb + 16 * MT <= nframes;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // one row at a time: without the barrier hipcc interleaves all the
            // rows and groups and spills the accumulators to scratch
            __builtin_amdgcn_sched_barrier(0);
            const int64_t f = fb + m * 16 + M::row(g, r);
            const bool f_in = rows_in || f < nframes;
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
                    const T lse = (mx + M::log_sum(sum)) + shift;
                    if (!LNO) {
                        const T inv = M::recip(sum);
#pragma unroll
                        for (int qq = 0; qq < GQ; ++qq)
#pragma unroll
                            for (int j = 0; j < 4; ++j) e[qq][j] *= inv;
                    }
                    if (f_in && writer[tq]) {
                        if (log_norm) log_norm[f * S + state_of[tq]] = lse;
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
                            const T lse = (mx + M::log_sum(e0 + e1)) + shift;
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
                                    if (log_norm) log_norm[f * S + kq + j0] = a0 + shift;
                                    llh_local += (double)(a0 + shift);
                                }
                                if (kq + j0 + 1 < K) {
                                    if (log_norm) log_norm[f * S + kq + j0 + 1] = a1 + shift;
                                    llh_local += (double)(a1 + shift);
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
        typedef unsigned int uint4_t __attribute__((ext_vector_type(4)));
        unsigned int* out = reinterpret_cast<unsigned int*>(resps);
        const int nblk = (K + kPackedComps - 1) / kPackedComps;
        const int64_t tiles = (nframes + kPackedFrames - 1) / kPackedFrames;
        // (a tile inside the frames and the components -- all but the last of a launch -- packs its
        // values as they are: the per-element `frame < nframes && k < K` was a 64-bit add, a 64-bit
        // compare and a select for each of a lane's 128 values)
        const bool inside = rows_in && kbase + 16 * NT <= K;
        auto pack_tile = [&](auto inside_t) {
            constexpr bool INSIDE = decltype(inside_t)::value;
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
                unsigned pw[2][2][kPackedPieces];              // [tile of the pair][word][piece]
#pragma unroll
                for (int mm = 0; mm < 2; ++mm) {
                    const int m = 2 * mp + mm;
                    const int64_t f0 = fb + m * 16 + M::row(g, 0);
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        v[r] = (INSIDE || (f0 + r < nframes && k < K)) ? (float)acc[m][nt][r] : 0.f;
                    split3(v[0], v[1], pw[mm][0]);
                    split3(v[2], v[3], pw[mm][1]);
                }
                // (the exchange is a cross-lane operation: every lane takes part, only the
                // store is conditional)
                uint4_t ch[kPackedPieces];
#pragma unroll
                for (int q = 0; q < kPackedPieces; ++q)
#pragma unroll
                    for (int wd = 0; wd < 2; ++wd) {
                        const auto sw = __builtin_amdgcn_permlane16_swap(pw[0][wd][q], pw[1][wd][q],
                                                                         false, false);
                        ch[q][wd] = sw[0]; ch[q][2 + wd] = sw[1];
                    }
                if (chunk_ok && k < nblk * kPackedComps) {
                    unsigned int* dst = out + packed_word(tau, nblk, k / kPackedComps,
                                                          k & (kPackedComps - 1), f6);
                    // non-temporal: streamed once; keeps the packed parameters in L2
#pragma unroll
                    for (int q = 0; q < kPackedPieces; ++q)
                        __builtin_nontemporal_store(
                            ch[q], reinterpret_cast<uint4_t*>(dst + q * kPackedPlaneWords));
                }
            }
        }
            };
        if (inside) pack_tile(std::true_type{});
        else pack_tile(std::false_type{});
    }
    if (llh_sum) {
        llh_local = wave_sum(llh_local);
        if (lane == 0) atomicAdd(llh_sum, llh_local);
    }
}

// The LNO epilogue of the float kernels on its own (mixture sets with G >= 4: a group
// is 4 values of a lane x GL lanes): the log-normalisers [T, S] and their sum, nothing
// else.  Everything that does not depend on the row -- the states of the lane's groups
// (an integer division each), which lanes write, the row pointers -- is taken out of
// the 16 MT x NT / 4 element loop, and the group width is a compile-time constant (the
// generic epilogue above spends 6000 instructions per tile of 32 x 256 logits on
// these, four times its arithmetic; the diagonal-covariance E-step was bound by it).
template <int NT, int MT, int GQ, int GL>
__device__ __forceinline__ void lognorm_epilogue(
    f32x4 (&acc)[MT][NT], int64_t fb, int64_t nframes, int kbase, int S, int G, int i, int g,
    int lane, float* __restrict__ log_norm, double* __restrict__ llh_sum, float shift) {
    using M = Mma<float>;
    constexpr int NG = NT / 4 / GQ;
    int state[NG];
    bool writes[NG];
#pragma unroll
    for (int tq = 0; tq < NG; ++tq) {
        state[tq] = (kbase + 64 * tq * GQ + 4 * (i & ~(GL - 1))) / G;
        writes[tq] = (i & (GL - 1)) == 0 && state[tq] < S;
        if (!writes[tq]) state[tq] = 0;
    }
    double llh_local = 0.0;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            __builtin_amdgcn_sched_barrier(0);
            const int64_t f = fb + m * 16 + M::row(g, r);
            const bool ok = f < nframes;
            float* row = log_norm + (ok ? f : 0) * S;
            float rowsum = 0.f;
#pragma unroll
            for (int tq = 0; tq < NG; ++tq) {
                float mx = acc[m][4 * tq * GQ][r];
#pragma unroll
                for (int c = 1; c < 4 * GQ; ++c) mx = M::mx(mx, acc[m][4 * tq * GQ + c][r]);
                mx = group_max(mx, GL);
                float sum = 0.f;
#pragma unroll
                for (int c = 0; c < 4 * GQ; ++c) sum += M::exp_neg(acc[m][4 * tq * GQ + c][r] - mx);
                sum = group_sum(sum, GL);
                const float lse = (mx + M::log_sum(sum)) + shift;
                if (ok && writes[tq]) {
                    if (log_norm) row[state[tq]] = lse;
                    rowsum += lse;
                }
            }
            llh_local += (double)rowsum;
        }
    }
    if (llh_sum) {
        llh_local = wave_sum(llh_local);
        if (lane == 0) atomicAdd(llh_sum, llh_local);
    }
}

// ... and for groups that live INSIDE a lane.  With NT = 16 column tiles a lane holds 16
// logits per frame row; when the packed image deals the components out lane-major (lane
// column i: component slots 16 i .. 16 i + 15 of the chunk, slot 16 i + c in tile c --
// packx_kernel's `lane_major`) a state's G <= 16 Gaussians are 16 / G runs of G registers
// of ONE lane: the log-sum-exp needs no cross-lane step at all (the layout above spends
// two DPP stages with their wait states per group and row, and a predicated store per
// group), the maxima come three at a time (v_max3), and the change of base folds into the
// subtraction: 2^(a L + nm) with nm = fl(-mx L) is 2^((a - mx) L) times 2^d, d = mx L + nm
// the EXACT rounding error of nm (one fma), taken out of the logarithm again.  Per logit:
// fma, exp, add (+ half a max) instead of max, sub, mul, exp, add and the shuffles.
template <int NT, int MT, int G>
__device__ __forceinline__ void lognorm_epilogue_lane_major(
    f32x4 (&acc)[MT][NT], int64_t fb, int64_t nframes, int kbase, int S, int i, int g, int lane,
    float* __restrict__ log_norm, double* __restrict__ llh_sum, float shift) {
    static_assert((NT == 16 || NT == 8) && (G == 4 || G == 8 || G == 16) && NT % G == 0,
                  "NT logits per lane and row, whole groups");
    constexpr int NG = NT / G;
    constexpr float L2E = 1.44269504088896340736f, LN2 = 0.69314718055994530942f;
    const int s0 = (kbase + NT * i) / G;                 // the lane's first state
    double llh_local = 0.0;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            __builtin_amdgcn_sched_barrier(0);           // (one row at a time: registers)
            const int64_t f = fb + m * 16 + 4 * g + r;
            const bool ok = f < nframes;
            float* row = log_norm + (ok ? f : 0) * S + s0;
            float rowsum = 0.f;
#pragma unroll
            for (int q = 0; q < NG; ++q) {
                float mx = acc[m][q * G][r];
#pragma unroll
                for (int c = 1; c + 1 < G; c += 2)
                    mx = __builtin_fmaxf(__builtin_fmaxf(mx, acc[m][q * G + c][r]), acc[m][q * G + c + 1][r]);
                mx = __builtin_fmaxf(mx, acc[m][q * G + G - 1][r]);
                const float nm = -mx * L2E;
                const float d = __builtin_fmaf(mx, L2E, nm);
                float sum = 0.f;
#pragma unroll
                for (int c = 0; c < G; ++c)
                    sum += (BEER_LNFI_TRANS_EVERY > 0 && c % (BEER_LNFI_TRANS_EVERY > 0 ? BEER_LNFI_TRANS_EVERY : 1) == 0)
                               ? __builtin_amdgcn_exp2f(__builtin_fmaf(acc[m][q * G + c][r], L2E, nm))
                               : exp2_valu(__builtin_fmaf(acc[m][q * G + c][r], L2E, nm));
                const float lse = __builtin_fmaf(__builtin_amdgcn_logf(sum) - d, LN2, mx) + shift;
                if (ok && s0 + q < S) {
                    if (log_norm) row[q] = lse;
                    rowsum += lse;
                }
            }
            llh_local += (double)rowsum;
        }
    }
    if (llh_sum) {
        llh_local = wave_sum(llh_local);
        if (lane == 0) atomicAdd(llh_sum, llh_local);
    }
}

}  // namespace beer_mfma
