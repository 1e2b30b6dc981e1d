// fp32 E-step on the fp16 matrix pipes ("f16x3").
//
// gfx950 runs v_mfma_f32_16x16x4_f32 at 1/16 of the rate of
// v_mfma_f32_16x16x32_f16 (MI355X_MICROARCH.md: 157 TF vs 2.5 PF dense).  Both
// GEMMs of the E-step are therefore evaluated with every fp32 operand split
// into two fp16 halves, v = hi + lo (hi = fp16(v), lo = fp16(v - hi): 22
// mantissa bits), and three fp16 MFMAs per product, accumulated in fp32:
//
//     a * b  ~=  a_hi b_hi + a_hi b_lo + a_lo b_hi        (|err| <= 2^-21 |a b|)
//
// i.e. the accuracy of an fp32 multiply (2^-24) to within a factor of 8, at a
// third of the fp16 rate = 5.3x the fp32 MFMA rate.  fp16's narrow exponent
// range is handled with exact power-of-two scalings: the frames are scaled so
// that |x| < 64 (products < 4096), every column of the packed parameter image
// so that its largest entry is below 2^14; both are undone in the epilogue.
// Entries more than 2^28 below the column maximum lose relative accuracy
// (fp16 subnormals) but contribute < 2^-38 of the column's dominant term.
//
// Same slab enumeration, component interleave and softmax epilogue as
// estep_mfma.hip; one k-step of the fp16 MFMA (32 deep) covers 8 slabs, lane
// k-block g (8 values) = slabs 8s+2g and 8s+2g+1.
//
// Operand mapping of v_mfma_f32_16x16x32_f16 (lane l: i = l & 15, g = l >> 4):
// A[i][k = 8g..8g+7], B[k = 8g..8g+7][n = i], C/D row 4g + r, column i.

#include <cstdlib>
#include <type_traits>

#include "estep_mfma.h"
#include "estep_tiles.h"

namespace beer_mfma {

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 hp2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

constexpr int kScaleBits = 7;          // |x * sx| < 2^7: products < 2^14
constexpr int kRespBits = 12;          // responsibilities are staged as r * 2^12 (fp16 range)
constexpr int kColBits = 14;           // column maximum of P < 2^14

// k-steps (8 slabs each), padded to an even count: the K1 loop is unrolled by two
__host__ __device__ inline int nk16_of(int cov, int D) {
    return ((nslab_of(cov, D) + 7) / 8 + 1) / 2 * 2;
}
// Row stride (floats) of K1's LDS frame tile: the D values, the constant 1 and at
// least 7 zeros (the padding slabs read them), with stride / 4 odd -- the 16 rows
// of an A-fragment ds_read_b128 then start in 16 different 16-byte slots of the
// 256-byte bank row (with stride 48 at D = 40 they fell on 4 slots: 74 % of the
// LDS cycles of the kernel were bank conflicts, profiles/r01_pmc.json).
__host__ __device__ inline int ld16_of(int D) {
    const int ld = 4 * d4_of(D) + 8;
    return (ld / 4) % 2 ? ld : ld + 4;
}
constexpr float kConstEps = 0.00048828125f;   // 2^-11: second constant of a frame row
constexpr int kPadBlocks = 16;         // look-ahead blocks behind the P image (a quarter k-step
                                       // of the second component half, see KS)

// Per-dimension frame scaling: sc[d] = s_d, a power of two with |x_d s_d| < 2^7,
// sc[64 + d] = 1 / s_d (D <= 64).  One scale per dimension rather than one for
// the whole matrix: features of very different magnitude (an energy next to
// cepstra, unnormalised filter-bank outputs) keep their own fp16 range.
// absmax[d] = bit pattern of max_t |x_td|: one wave reads one frame per step
// (lane = dimension), no atomics until the end.
__global__ __launch_bounds__(512) void absmax_kernel(const float* __restrict__ X, int64_t nframes,
                                                     int D, unsigned* __restrict__ out) {
    __shared__ float red[8][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    float m = 0.f;
    if (lane < D) {
        int64_t f = (int64_t)blockIdx.x * nwave + wave;
        const int64_t stride = (int64_t)gridDim.x * nwave;
        // four frames in flight per wave
        for (; f + 3 * stride < nframes; f += 4 * stride) {
            const float a = fabsf(X[f * D + lane]), b = fabsf(X[(f + stride) * D + lane]);
            const float c = fabsf(X[(f + 2 * stride) * D + lane]);
            const float d = fabsf(X[(f + 3 * stride) * D + lane]);
            const float ab = a > b ? a : b, cd = c > d ? c : d, q = ab > cd ? ab : cd;
            m = q > m ? q : m;                                   // NaN never wins
        }
        for (; f < nframes; f += stride) {
            const float a = fabsf(X[f * D + lane]);
            m = a > m ? a : m;
        }
    }
    red[wave][lane] = m;
    __syncthreads();
    if (wave == 0 && lane < D) {
        for (int w = 1; w < nwave; ++w) m = red[w][lane] > m ? red[w][lane] : m;
        atomicMax(out + lane, __float_as_uint(m));
    }
}

__global__ void scale_kernel(const unsigned* __restrict__ absmax, int D, float* __restrict__ sc) {
    const int d = threadIdx.x;
    if (d >= 64) return;
    float sx = 1.f;
    if (d < D) {
        const float m = __uint_as_float(absmax[d]);
        if (m > 0.f && m < 3.0e38f) {
            int e;
            frexpf(m, &e);                                       // m < 2^e
            sx = ldexpf(1.f, kScaleBits - e);
        }
    }
    sc[d] = sx;
    sc[64 + d] = 1.f / sx;
}

// sum_t |x_td| per dimension (same access pattern as absmax_kernel)
__global__ __launch_bounds__(512) void abssum_kernel(const float* __restrict__ X, int64_t nframes,
                                                     int D, double* __restrict__ out) {
    __shared__ double red[8][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    double m = 0.0;
    if (lane < D)
        for (int64_t f = (int64_t)blockIdx.x * nwave + wave; f < nframes;
             f += (int64_t)gridDim.x * nwave) {
            const float a = fabsf(X[f * D + lane]);
            if (a == a && a < 3.0e38f) m += (double)a;
        }
    red[wave][lane] = m;
    __syncthreads();
    if (wave == 0 && lane < D) {
        for (int w = 1; w < nwave; ++w) m += red[w][lane];
        atomicAdd(out + lane, m);
    }
}

// hazard = 1 when some dimension's largest magnitude is more than 2^kRangeBits
// times its mean magnitude: after scaling the maximum to 2^7, the products of
// typical values would fall into fp16's subnormals (< 2^-14) and lose bits.
constexpr int kRangeBits = 9;
__global__ void hazard_kernel(const unsigned* __restrict__ absmax, const double* __restrict__ abssum,
                              int64_t nframes, int D, int* __restrict__ hazard) {
    int bad = 0;
    for (int d = 0; d < D; ++d) {
        const double mx = (double)__uint_as_float(absmax[d]);
        const double mean = abssum[d] / (double)(nframes > 0 ? nframes : 1);
        if (!(mx < 3.0e38)) bad = 1;                               // inf / huge values
        if (mean > 0.0 && mx > ldexp(mean, kRangeBits)) bad = 1;
    }
    *hazard = bad;
}

int launch_scales(const float* X, int64_t nframes, int D, unsigned* absmax, float* sc,
                  hipStream_t s) {
    hipError_t e = hipMemsetAsync(absmax, 0, 64 * sizeof(unsigned), s);
    if (e != hipSuccess) return -(int)e;
    int64_t blocks = (nframes + 8 * 16 - 1) / (8 * 16);
    if (blocks > 512) blocks = 512;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(512), 0, s, X, nframes, D,
                       absmax);
    hipLaunchKernelGGL(scale_kernel, dim3(1), dim3(64), 0, s, absmax, D, sc);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

// Value of contraction entry (slab, e) for component k (the logic of
// pack_kernel in estep_mfma.hip) times the inverse frame scaling.
__device__ inline double entry_value(int cov, int D, int K, int k, int slab, int e,
                                     const float* __restrict__ E, const float* __restrict__ logw,
                                     const float* __restrict__ isx, bool* is_const,
                                     const float* row_in = nullptr) {
    const int D4 = d4_of(D), Dp = 4 * D4, nslab = nslab_of(cov, D), Q = stats_dim(cov, D);
    *is_const = false;
    if (slab >= nslab) return 0.0;
    const int t = slab_entry(cov, D, slab);
    const int a = t & 0xff, b = ((t >> 8) & 0xff) + e, sq = t >> 16;
    const bool constant = (a == Dp && b - e == Dp);
    if (constant) *is_const = (e == 0);
    if (k >= K) return 0.0;
    const float* row = row_in ? row_in : E + (size_t)k * Q;
    if (sq)
        return b < D ? -0.5 * (double)row[cov == BEER_ISO ? D : D + b] * (double)isx[b] * (double)isx[b]
                     : 0.0;
    if (a < D) {
        if (b >= D || b < a) return 0.0;
        return (b == a ? -0.5 * (double)row[D + a * D + a]
                       : -0.5 * ((double)row[D + a * D + b] + (double)row[D + b * D + a])) *
               (double)isx[a] * (double)isx[b];
    }
    if (!constant) return b < D ? (double)row[b] * (double)isx[b] : 0.0;
    if (e != 0) return 0.0;
    const double zero = cov == BEER_ISO ? 0.5 * (double)D : 0.5;
    return -0.5 * (double)row[Q - 2] + zero * (double)row[Q - 1] - 0.5 * (double)D * kLog2Pi +
           (logw ? (double)logw[k] : 0.0);
}

// dynamic LDS of pack16_kernel: [moments: Sigma (D*D, or D variances), mu (D), 2 D of
// elimination scratch] as doubles, then the component's row of E as floats
__host__ __device__ inline size_t pack16_moment_bytes(int cov, int D) {
    return (size_t)((cov == BEER_FULL ? D * D : D) + 3 * D) * sizeof(double);
}

// One workgroup per (padded) component: column scale, then its fp16 hi / lo
// images at P16[chunk][kstep][tile][hi 64 x 8 | lo 64 x 8] halves, and 1 / scale.
// Component SLOT blockIdx.x of the image is component (slot / Gp) * G + slot % Gp when
// slot % Gp < G, else padding (G <= Gp: the groups of a mixture set padded to a power
// of two, so that any number of components per state runs on the group-aligned
// kernels; G == Gp: slots are components).
__global__ __launch_bounds__(1024) void pack16_kernel(int cov, int D, int K, int NT, const float* __restrict__ E,
                              const float* __restrict__ logw, const float* __restrict__ sc,
                              _Float16* __restrict__ P, float* __restrict__ inv_scale,
                              int* __restrict__ tab, int G = 1, int Gp = 1,
                              const float* __restrict__ moments = nullptr) {
    __shared__ double red[16];
    extern __shared__ __attribute__((aligned(16))) char pack_lds[];
    const int nk = nk16_of(cov, D), nent = nk * 32;
    const int slot = blockIdx.x;
    const int k = slot % Gp < G ? (slot / Gp) * G + slot % Gp : K;     // K: a padded slot
    const int chunk = slot / (NT * 16), kk = slot % (NT * 16);
    const int c = 4 * (kk / 64) + (kk % 4), i = (kk % 64) / 4;
    const float* isx = sc + 64;
    if (slot == 0)
        for (int s = threadIdx.x; s < (nk + 1) * 8; s += blockDim.x) {
            // padding slabs read the zero columns behind the constants of a frame row
            const int Dp = 4 * d4_of(D);
            tab[s] = s < nslab_of(cov, D) ? slab_entry(cov, D, s) : ((Dp + 1) | ((Dp + 4) << 8));
        }
    // the component's expected statistics, staged once (the entries are gathered from
    // all over the row: from global memory every gather was a dependent L2 round trip)
    const int Qs = stats_dim(cov, D);
    float* rowl = reinterpret_cast<float*>(pack_lds + pack16_moment_bytes(cov, D));
    if (k < K)
        for (int q = threadIdx.x; q < Qs; q += blockDim.x) rowl[q] = E[(size_t)k * Qs + q];
    __syncthreads();
    double mx = 0.0;
    bool dummy;
    for (int q = threadIdx.x; q < nent; q += blockDim.x) {
        const double v = fabs(entry_value(cov, D, K, k, q / 4, q % 4, E, logw, isx, &dummy, rowl));
        mx = v > mx ? v : mx;
    }
    mx = block_max(mx, red);
    double scale = 1.0;
    if (mx > 0.0) {
        int e;
        frexp(mx, &e);                                       // mx < 2^e
        scale = ldexp(1.0, kColBits - e);
    }
    if (threadIdx.x == 0) inv_scale[slot] = k < K ? (float)(1.0 / scale) : 1.0e30f;

    // The component's own first and second moments, for the compensation below:
    // mu = E[Lambda]^-1 E[Lambda mu], Sigma = E[Lambda]^-1, from the expected
    // statistics themselves (E = [Lambda mu, -Lambda / 2, ...]).
    const int Dp = 4 * d4_of(D), Q = stats_dim(cov, D);
    double* S2 = reinterpret_cast<double*>(pack_lds);        // full: Sigma [D, D]; else var [D]
    double* mu = S2 + (cov == BEER_FULL ? D * D : D);
    double* cr = mu + D;                                     // 2 D (spd_inverse scratch)
    if (k < K && moments && cov == BEER_FULL) {
        // ... or as the caller has them from the M-step (beer_nw_update): no inverse here
        const float* mk = moments + (size_t)k * (D + D * D);
        for (int a = threadIdx.x; a < D; a += blockDim.x) mu[a] = (double)mk[a];
        for (int idx = threadIdx.x; idx < D * D; idx += blockDim.x) S2[idx] = (double)mk[D + idx];
    } else if (k < K) {
        const float* row = rowl;
        if (cov == BEER_FULL) {
            for (int idx = threadIdx.x; idx < D * D; idx += blockDim.x) {
                const int a = idx / D, b = idx - a * D;
                S2[idx] = -((double)row[D + a * D + b] + (double)row[D + b * D + a]);
            }
            spd_inverse(S2, D, cr, true);
            for (int a = threadIdx.x; a < D; a += blockDim.x) {
                double m = 0.0;
                for (int b = 0; b < D; ++b) m += S2[a * D + b] * (double)row[b];
                mu[a] = m;
            }
        } else {
            for (int b = threadIdx.x; b < D; b += blockDim.x) {
                const double lam = -2.0 * (double)row[cov == BEER_ISO ? D : D + b];
                S2[b] = 1.0 / lam;
                mu[b] = (double)row[b] / lam;
            }
        }
    }
    __syncthreads();
    // E[phi~_q(x s)] of entry q under N(mu, Sigma): what the entry multiplies on average
    // over the frames the component explains (s = the frame scales)
    auto expect = [&](int slab, int e) -> double {
        if (slab >= nslab_of(cov, D)) return 0.0;
        const int t = slab_entry(cov, D, slab);
        const int a = t & 0xff, b = ((t >> 8) & 0xff) + e, sq = t >> 16;
        if (sq) return b < D ? (mu[b] * mu[b] + S2[b]) * (double)sc[b] * (double)sc[b] : 0.0;
        if (a < D) {
            if (b >= D || b < a) return 0.0;
            return (mu[a] * mu[b] + S2[a * D + b]) * (double)sc[a] * (double)sc[b];
        }
        if (a == Dp && b - e == Dp) return 0.0;              // the constant: handled below
        return b < D ? mu[b] * (double)sc[b] : 0.0;
    };
    _Float16* base = P + ((size_t)chunk * nk * NT) * 1024;
    auto put = [&](int q, _Float16 hi, _Float16 lo) {
        const int s = q / 32, g = (q % 32) / 8, j = q % 8;
        _Float16* dst = base + ((size_t)s * NT + c) * 1024 + (g * 16 + i) * 8 + j;
        dst[0] = hi;
        dst[512] = lo;
    };
    double bias = 0.0, cs = 0.0;
    int qc = -1;                                              // the constant's entry (e = 0)
    for (int q = threadIdx.x; q < nent; q += blockDim.x) {
        bool is_const;
        double v = entry_value(cov, D, K, k, q / 4, q % 4, E, logw, isx, &is_const, rowl) * scale;
        if (k >= K) v = is_const ? -1.0 : 0.0;            // padded component: logit -1e30
        const float vf = (float)v;
        const _Float16 hi = (_Float16)vf;
        const _Float16 lo = (_Float16)(vf - (float)hi);
        // (the constant and its remainder entry are written once, below)
        bool const_slab = false;
        if (k < K && q / 4 < nslab_of(cov, D)) {
            const int t = slab_entry(cov, D, q / 4);
            const_slab = (t & 0xff) == Dp && ((t >> 8) & 0xff) == Dp && (t >> 16) == 0;
        }
        if (!(const_slab && q % 4 < 2)) put(q, hi, lo);
        if (is_const) { qc = q; cs = v; }
        else if (k < K && v != 0.0) bias += ((double)hi + (double)lo - v) * expect(q / 4, q % 4);
    }
    // Compensation of the image's own rounding.  An entry carries 22 bits; its error is
    // the same for every frame, so over the frames of the component the logit is off by
    // sum_q err_q E[phi_q] -- a bias that does not average out in the statistics
    // (measured 1e-5 .. 7e-5 of the counts).  It is known here: subtract it from the
    // constant, which gets 22 more bits for the purpose -- the frame rows carry 2^-11
    // next to their 1, and the constant slab's second entry holds the remainder * 2^11.
    bias = block_sum(bias, red);
    if (qc >= 0 && k < K) {
        const double want = (bias == bias && fabs(bias) < 1.0) ? cs - bias : cs;
        const float vf = (float)want;
        const _Float16 hi = (_Float16)vf;
        const _Float16 lo = (_Float16)((float)(want - (double)hi));
        const double rem = (want - (double)hi - (double)lo) * (1.0 / (double)kConstEps);
        const bool ok = want == want && fabs(want) < 65000.0;        // (-inf weight: as is)
        const float rf = ok ? (float)rem : 0.f;
        const _Float16 h1 = (_Float16)rf;
        put(qc, hi, lo);
        put(qc + 1, h1, (_Float16)(rf - (float)h1));
    }
}

// One workgroup per component; full covariance inverts a D x D matrix in it, which is
// pure latency: 1024 threads while there is about one matrix per CU (cf. nw_threads)
inline int pack16_threads(int cov, int D, int slots) {
    return cov == BEER_FULL && D >= 16 && slots <= 512 ? 1024 : 256;
}
inline size_t pack16_lds(int cov, int D) {
    return pack16_moment_bytes(cov, D) + (size_t)stats_dim(cov, D) * sizeof(float);
}

// v = hi + lo with hi = fp16(v), lo = fp16(v - hi), both round-to-nearest
// (v_cvt_pk_f16_f32): v - hi is exact in fp32, so |v - hi - lo| <= 2^-22 |v|
// unless lo falls into the fp16 subnormals (|v| < 2^-3: error <= 2^-25).
// lo comes from two v_fma_mix: fma(f32(hi half), -1, v) rounded to fp16 into the
// low / high half of the destination -- 3 instructions per pair instead of 5
// (hipcc's own choice: convert hi back, packed subtract, convert).
__device__ __forceinline__ void split2(float a, float b, hp2& hi, hp2& lo) {
    const f32x2 v = {a, b};
    hi = __builtin_convertvector(v, hp2);
    const unsigned h = __builtin_bit_cast(unsigned, hi);
    unsigned l;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "v"(a));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "+v"(l) : "v"(h), "v"(b));
    lo = __builtin_bit_cast(hp2, l);
}

__device__ __forceinline__ void split8(const f32x4& p0, const f32x4& p1, h8& hi, h8& lo) {
    union { h8 v; hp2 p[4]; } H, L;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float a = q < 2 ? p0[2 * q] : p1[2 * q - 4];
        const float b = q < 2 ? p0[2 * q + 1] : p1[2 * q - 3];
        split2(a, b, H.p[q], L.p[q]);
    }
    hi = H.v;
    lo = L.v;
}

// ---------------------------------------------------------------------------
// K1 on the fp16 pipes: one wave owns 16 MT frames x 16 NT components.
// KS = 2 (one mixture, packed output): waves 2p, 2p + 1 share the 16 MT frames of
// pair p and own 16 NT components each of the chunk's 32 NT -- every wave
// streams half of the packed parameters for twice the frames (the B stream
// through the vector L1, 29 GB per launch at K = 256 with one wave per frame
// tile, is what bounds the main loop), and the softmax spans the two waves
// (softmax_epilogue_pair).
// ---------------------------------------------------------------------------
// SQ = false: no "square" slabs in the table (full covariance), the per-product
// select between x_j^2 and x_a x_j drops out of the A-fragment arithmetic.
// Several component chunks over the same frames: a 1-D grid whose blocks are dealt to
// the 8 XCDs round-robin, laid out so that the chunk blocks of one frame block are
// neighbours on ONE XCD (they read the same frames at about the same time: one L2
// miss, the others hit -- with a (frames, chunks) grid every chunk pass streamed the
// frames from HBM again).  The chunks are taken `cg` at a time (all frame blocks for
// chunks 0 .. cg-1, then the next cg): the packed parameters of the cg chunks an XCD
// works on must stay in its 4 MiB L2 -- with all 8 full-covariance chunks of config 3
// interleaved (7.6 MB) the parameter stream missed L2 instead of the frames.
// Grid size xcd_grid(nx, ny, cg); false = padding block.
inline int xcd_chunk_group(int ny, size_t chunk_bytes) {
    int cg = (int)((size_t)(2 << 20) / (chunk_bytes ? chunk_bytes : 1));
    return cg < 1 ? 1 : (cg > ny ? ny : cg);
}
inline unsigned xcd_grid(int64_t nx, int ny, int cg) {
    return (unsigned)((nx + 7) / 8 * 8 * cg * ((ny + cg - 1) / cg));
}
__device__ inline bool xcd_block(int64_t nx, int ny, int cg, int64_t& bx, int& by) {
    const unsigned per_group = (unsigned)((nx + 7) / 8 * 8 * cg);
    const unsigned grp = blockIdx.x / per_group, rem = blockIdx.x - grp * per_group;
    const unsigned xcd = rem & 7, slot = rem >> 3;
    by = (int)(grp * cg + slot % (unsigned)cg);
    bx = (int64_t)(slot / (unsigned)cg) * 8 + xcd;
    return bx < nx && by < ny;
}

template <int NT, int MT, int GQ, bool PACKED, int KS = 1, bool SQ = true, bool LNO = false,
          bool SETS = false>
__global__ __launch_bounds__(kThreads, MT * NT <= 32 ? 2 : 1) void llh16_kernel(
    int64_t nframes, int D, int K, int S, int G, int gl, int jw, int nk,
    const float* __restrict__ X, const _Float16* __restrict__ Pall,
    const float* __restrict__ inv_scale, const float* __restrict__ sc,
    const int* __restrict__ tab, float* __restrict__ resps, float* __restrict__ log_norm,
    double* __restrict__ llh_sum, float* __restrict__ xt_out, int xt_floats, int nku, int cg) {
    using acc_t = f32x4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int D4 = d4_of(D), Dp = 4 * D4, LD = ld16_of(D);    // 16-byte aligned rows
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    constexpr int FW = 16 * MT, NGRP = kThreads / 64 / KS;      // frame groups per workgroup
    // KS = 2 without PACKED: mixture sets whose groups (G <= 128 components) lie
    // inside one wave's half of the chunk -- no cross-wave softmax needed
    const int grp = wave / KS, part = wave % KS;
    float* xw = reinterpret_cast<float*>(smem) + grp * (FW * LD);
    int* tabs = reinterpret_cast<int*>(reinterpret_cast<float*>(smem) + NGRP * FW * LD);
    int64_t bx = blockIdx.x;
    int by = 0;
    {
        const int nch = (K + 16 * NT * KS - 1) / (16 * NT * KS);
        constexpr int FBK = FW * NGRP;
        if (nch > 1 && !xcd_block((nframes + FBK - 1) / FBK, nch, cg, bx, by)) return;
    }
    const int64_t fb = (bx * NGRP + grp) * FW;
    for (int idx = tid; idx < (nk + 1) * 8; idx += kThreads) tabs[idx] = tab[idx];

    if ((D & 3) == 0) {
        // rows of whole float4: 16-byte loads and LDS stores, one division per 4 values
        const int C4 = D >> 2;
        const f32x4* X4 = reinterpret_cast<const f32x4*>(X);
        const f32x4* sc4 = reinterpret_cast<const f32x4*>(sc);
        static_assert(64 * KS == 2 * FW, "two lanes per frame row");
        if ((C4 & 1) == 0 && C4 <= 16) {
            // a lane = (row, half of the row): source, scales and destination are one
            // per-lane base plus a constant per piece (no division, no address per piece)
            const int lg = part * 64 + lane, r = lg & (FW - 1), h = lg / FW;
            const int64_t f = fb + r;
            const bool valid = f < nframes;
            const f32x4* src = X4 + (valid ? f : nframes - 1) * C4 + h;
            float* dst = xw + r * LD + 4 * h;
            // (all loads first, on clamped piece numbers: a load behind a branch is
            // issued and waited for on its own)
            const int np = C4 >> 1;
            f32x4 xv[8];
#pragma unroll
            for (int it = 0; it < 8; ++it) xv[it] = src[2 * (it < np ? it : np - 1)];
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                if (it >= np) break;
                f32x4 v = xv[it] * sc4[h + 2 * it];
                if (!valid) v = f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(dst + 8 * it) = v;
            }
        } else
        for (int idx = part * 64 + lane; idx < FW * C4; idx += 64 * KS) {
            const int r = idx / C4, c4 = idx - r * C4;
            const int64_t f = fb + r;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (f < nframes) v = X4[f * C4 + c4] * sc4[c4];
            *reinterpret_cast<f32x4*>(xw + r * LD + 4 * c4) = v;
        }
        // columns D .. LD - 1: the constant 1 at Dp (= D), zeros behind it
        for (int idx = part * 64 + lane; idx < FW * 2; idx += 64 * KS) {
            const int r = idx >> 1, h = idx & 1;
            *reinterpret_cast<f32x4*>(xw + r * LD + D + 4 * h) =
                f32x4{h == 0 ? 1.f : 0.f, h == 0 ? kConstEps : 0.f, 0.f, 0.f};
        }
    } else {
        for (int idx = part * 64 + lane; idx < FW * LD; idx += 64 * KS) {
            const int r = idx / LD, c = idx - r * LD;
            const int64_t f = fb + r;
            float v = 0.f;
            if (c < D) { if (f < nframes) v = X[f * D + c] * sc[c]; }
            else if (c == Dp) v = 1.f;
            else if (c == Dp + 1) v = kConstEps;
            xw[idx] = v;
        }
    }
    __syncthreads();
    if constexpr (KS == 2) {
        // The group's 64 scaled frames are one tile of the accumulation kernel:
        // leave them behind transposed, [D + 2][kA16XS] (rows D, D + 1 = 1, 0; see
        // xt_image_kernel, which this replaces -- one pass over the frames less).
        static_assert(KS != 2 || FW == 64, "a frame group is one 64-frame tile");
        if (xt_out && fb < (nframes + 63) / 64 * 64) {
            float* img = xt_out + (fb / 64) * (size_t)xt_floats;
            constexpr int XS = 68, C4 = XS / 4;
            for (int e4 = part * 64 + lane; e4 < xt_floats / 4; e4 += 64 * KS) {
                const int row = e4 / C4, c = 4 * (e4 - row * C4);
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (row < D) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = c + j < 64 ? xw[(c + j) * LD + row] : 0.f;
                } else if (row == D) {
                    v = f32x4{1.f, 1.f, 1.f, 1.f};
                }
                __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(img + 4 * e4));
            }
        }
    }

    acc_t acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int c = 0; c < NT; ++c) acc[m][c] = acc_t{0, 0, 0, 0};

    const float* xrow[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) xrow[m] = xw + (m * 16 + i) * LD;

    const int kbase = by * (16 * NT * KS) + part * (16 * NT);
    // mixture sets: a wave whose half of the last chunk lies behind the last component
    // has staged its share of the frames and is done (no barrier below for it)
    if (KS == 2 && (SETS || !PACKED) && kbase >= K) return;
    // the B stream of this chunk is one linear sequence of (k-step, tile) blocks
    // of 2 KiB = 128 u4 (hi 64 lanes x 16 B, lo 64 lanes x 16 B); with KS = 2 a
    // k-step has 2 NT tiles and this wave reads tiles part * NT ..
    const u4* Pl = reinterpret_cast<const u4*>(Pall + (size_t)by * nk * (NT * KS) * 1024) +
                   lane + (size_t)part * NT * 128;
    const int* tl = tabs + 2 * g;

    // A fragments as 32-bit words (two fp16 each): word w of tile m holds the
    // entries 2w, 2w+1 of the lane's 8-deep k-block (w < 2: first slab).
    struct AFrag { unsigned hi[MT][4], lo[MT][4]; };
    auto frag = [](const unsigned (&w)[4]) {
        return __builtin_bit_cast(h8, u4{w[0], w[1], w[2], w[3]});
    };
    // one slab (half a k-block) of tile m of k-step s: LDS reads, 4 products,
    // fp16 split -> words 2h, 2h+1
    auto make_half = [&](int s, int m, int h, AFrag& f) {
        const int t = tl[8 * s + h];
        const int a = t & 0xff, j = (t >> 8) & 0xff;
        const bool sq = SQ && (t >> 16) != 0;
        const f32x4 bb = *reinterpret_cast<const f32x4*>(xrow[m] + j);
        const float xx = xrow[m][a];
        f32x4 p;
#pragma unroll
        for (int e = 0; e < 4; ++e) p[e] = bb[e] * (sq ? bb[e] : xx);     // v_cndmask, no branch
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            hp2 hh, ll;
            split2(p[2 * e], p[2 * e + 1], hh, ll);
            f.hi[m][2 * h + e] = __builtin_bit_cast(unsigned, hh);
            f.lo[m][2 * h + e] = __builtin_bit_cast(unsigned, ll);
        }
    };
    // Software pipeline in "quarters" of a k-step (QT = NT / 4 column tiles):
    // the B fragments of the next quarter are loaded before the MFMA block of
    // the current one (sched_group_barrier pins them there: hipcc otherwise
    // sinks every load next to its first use), and a slice of the A fragments
    // of the next k-step is prepared in between.
    constexpr int QT = NT / 4;
    struct BFrag { h8 hi[QT], lo[QT]; };
    auto load_b = [&](int64_t blk, BFrag& b) {               // blocks blk .. blk + QT - 1
#pragma unroll
        for (int c = 0; c < QT; ++c) {
            b.hi[c] = __builtin_bit_cast(h8, Pl[(size_t)(blk + c) * 128]);
            b.lo[c] = __builtin_bit_cast(h8, Pl[(size_t)(blk + c) * 128 + 64]);
        }
    };
    auto quarter = [&](int s, int q, const AFrag& cur, AFrag& nxt, const BFrag& b, BFrag& bn) {
        // P is padded by one quarter (q + 1 = 4: first quarter of the next k-step)
        if constexpr (KS == 1) load_b((int64_t)s * NT + (q + 1) * QT, bn);
        else load_b((int64_t)(s + (q + 1) / 4) * (NT * KS) + ((q + 1) % 4) * QT, bn);
        // slices of the next A: MT * 2 halves over 4 quarters
        // (half-major: the halves of one quarter share the table entry)
#pragma unroll
        for (int hh = q * MT * 2 / 4; hh < (q + 1) * MT * 2 / 4; ++hh)
            make_half(s + 1, hh % MT, hh / MT, nxt);         // the table is padded by one k-step
#pragma unroll
        for (int c = 0; c < QT; ++c) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
                acc[m][q * QT + c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                    frag(cur.hi[m]), b.hi[c], acc[m][q * QT + c], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MT; ++m)
                acc[m][q * QT + c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                    frag(cur.hi[m]), b.lo[c], acc[m][q * QT + c], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MT; ++m)
                acc[m][q * QT + c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                    frag(cur.lo[m]), b.hi[c], acc[m][q * QT + c], 0, 0, 0);
        }
#if BEER_K1_SCHED == 1
        // the quarter's LDS reads first, their arithmetic two thirds of an MFMA block
        // later (an LDS read waited for at once stalls the wave ~100 cycles, 8 times a
        // k-step), the arithmetic itself in the issue gaps of the MFMAs
        __builtin_amdgcn_sched_group_barrier(0x020, 2 * QT, 0);          // VMEM reads
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);               // DS reads
        __builtin_amdgcn_sched_group_barrier(0x008, MT * QT, 0);         // MFMA
#pragma unroll
        for (int r = 0; r < 2 * MT * QT; ++r) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);           // MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);           // VALU
        }
#else
        __builtin_amdgcn_sched_group_barrier(0x020, 2 * QT, 0);          // VMEM reads
        __builtin_amdgcn_sched_group_barrier(0x100, MT, 0);              // DS reads
        __builtin_amdgcn_sched_group_barrier(0x008, 3 * MT * QT, 0);     // MFMA
#endif
    };
    auto kstep = [&](int s, const AFrag& cur, AFrag& nxt, BFrag& b0, BFrag& b1) {
        quarter(s, 0, cur, nxt, b0, b1);
        quarter(s, 1, cur, nxt, b1, b0);
        quarter(s, 2, cur, nxt, b0, b1);
        quarter(s, 3, cur, nxt, b1, b0);
    };
    AFrag f0, f1;
    BFrag b0, b1;
#pragma unroll
    for (int hh = 0; hh < MT * 2; ++hh) make_half(0, hh % MT, hh / MT, f0);
    load_b(0, b0);
    // (the image is padded to an even number of k-steps; only those that hold slabs run)
    for (int s = 0; s < nku; s += 2) {
        kstep(s, f0, f1, b0, b1);
        if (s + 1 < nku) kstep(s + 1, f1, f0, b0, b1);
    }

    // undo the column scaling: column (tile c, lane-column i) is component
    // kbase + 64 (c / 4) + 4 i + c % 4
#pragma unroll
    for (int q = 0; q < NT / 4; ++q) {
        const f32x4 inv = *reinterpret_cast<const f32x4*>(inv_scale + kbase + 64 * q + 4 * i);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m][4 * q + j] *= inv[j];
    }
    // SETS: packed output of a mixture SET (groups inside one wave's components)
    if constexpr (KS == 2 && PACKED && !SETS) {
        float* xch = reinterpret_cast<float*>(tabs + (nk + 1) * 8);
        softmax_epilogue_pair<NT, MT>(acc, fb, nframes, kbase, K, i, g, lane, wave, xch, resps,
                                      log_norm, llh_sum);
    } else {
        softmax_epilogue<float, NT, MT, GQ, PACKED, LNO>(acc, fb, nframes, kbase, K, S, G, gl, jw, i,
                                                         g, lane, resps, log_norm, llh_sum);
    }
}

// covariance type of the E-step being launched (the launch helpers below take the
// shape, not the type; SQ = false kernels are full covariance by construction)
thread_local int g_cov_of_launch = BEER_FULL;

template <int NT, int MT, int GQ, bool PACKED = false, int KS = 1, bool SQ = true, bool LNO = false,
          bool SETS = false>
int launch_llh16(int64_t nframes, int D, int K, int S, int G, int gl, int jw, int nchunks, int nk,
                 const float* X, const _Float16* P, const float* inv_scale, const float* sc,
                 const int* tab, float* resps, float* log_norm, double* llh_sum, hipStream_t s,
                 float* xt_out = nullptr, int xt_floats = 0) {
    const int LD = ld16_of(D);
    // k-steps that hold slabs: the slab count is the table's (full: SQ = false)
    const int nku = (nslab_of(SQ ? g_cov_of_launch : BEER_FULL, D) + 7) / 8;
    constexpr int FB = 16 * MT * (kThreads / 64) / KS;
    const size_t lds = (size_t)FB * LD * sizeof(float) + (size_t)(nk + 1) * 8 * sizeof(int) +
                       (KS == 2 ? 8 * 16 * MT * sizeof(float) : 0);
    const int64_t blocks = (nframes + FB - 1) / FB;
    if (nchunks != (K + 16 * NT * KS - 1) / (16 * NT * KS)) return BEER_EINVAL;
    const int cg = xcd_chunk_group(nchunks, (size_t)nku * NT * KS * 2048);
    hipLaunchKernelGGL((llh16_kernel<NT, MT, GQ, PACKED, KS, SQ, LNO, SETS>),
                       dim3(nchunks > 1 ? xcd_grid(blocks, nchunks, cg) : (unsigned)blocks),
                       dim3(kThreads), lds, s, nframes, D, K, S, G, gl, jw, nk, X, P, inv_scale,
                       sc, tab, resps, log_norm, llh_sum, xt_out, xt_floats, nku, cg);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

// ---------------------------------------------------------------------------
// K2 on the fp16 pipes: S[k, q] += sum_t r[t,k] * PHI_q(x_t) with frames as the
// contraction index (32 per MFMA).  Workgroup = 4 waves (one per SIMD, so that a
// wave can hold a 128 x 64 output tile twice) sharing one LDS tile of 64 frames:
// R^T [128 components][64 frames] split into fp16 hi / lo when it is staged (each
// element once, reused by every statistic tile), X^T [D + 2 rows][64 frames] in
// fp32 (rows D, D+1 = the constants 1, 0).  A wave owns 128 components x NQ
// statistic tiles: A fragments = one ds_read_b128 of R^T hi / lo per component
// tile, B fragments = products of two X^T rows over the lane's 8 frames, split
// on the fly and reused by the 8 component tiles (LDS traffic per MFMA is what
// bounds this kernel: 32 ds_read_b128 per 96 MFMAs).  A workgroup sums at most
// kA16MaxFrames frames in fp32 -- one rounding per 32-frame MFMA, 384 per sum,
// against 4096 for the same frames on the 4-deep fp32 MFMA -- and adds its
// partial sums to the fp64 image with atomics.
// ---------------------------------------------------------------------------
constexpr int kA16Threads = 256;     // 4 waves, one per SIMD: 512 registers per lane
constexpr int kA16MC = 8;            // component tiles per workgroup (128 components)
constexpr int kA16MCsr = 4;          // ... with state responsibilities AND 4 statistic tiles
constexpr int kA16FT = 64;           // frames per LDS tile (2 k-steps)
constexpr int kA16XS = kA16FT + 4;   // X^T row stride (floats), 16-byte aligned
constexpr int kA16RS = kA16FT;       // R^T row = 8 chunks of 8 frames (16 B), chunk c of row
                                     // r stored at position c ^ (r & 7): conflict-free
                                     // ds_read_b128 for the MFMA lane groups, 2-way ds_write_b32
constexpr int kA16MaxFrames = 16384; // frames per workgroup: 384 fp32 roundings per sum

constexpr int kPiece = kA16Threads * 16;          // bytes one load of the workgroup moves
// The packed responsibilities start with the frame scales K1 computed (64 scales,
// 64 inverses): K2 reuses them instead of a second pass over the frames.
constexpr int kPackedHeader = 128 * sizeof(float);

inline int xt_rows(int D) { return D + 2; }                              // + ones, zeros
inline int xt_pieces(int D) { return (xt_rows(D) * kA16XS * 4 + kPiece - 1) / kPiece; }


template <int NQ, bool HAS_SR>
__global__ __launch_bounds__(kA16Threads, 1) void acc16_kernel(
    int64_t nframes, int D, int K, int G, int S, int nslab, const float* __restrict__ X,
    const float* __restrict__ R, const float* __restrict__ SR, const int* __restrict__ tab,
    const float* __restrict__ sc, int64_t frames_per_block, double* __restrict__ Sp, int gx,
    int gy, int gz) {
    // with state responsibilities the staging registers double: with 4 statistic
    // tiles per wave only half the component tile keeps the kernel free of spills
    constexpr int MC = HAS_SR && NQ > 2 ? kA16MCsr : kA16MC;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int D4 = d4_of(D), Dp = 4 * D4, nq = nslab * 4;
    // XCD-aware mapping: workgroups are dealt round-robin to the 8 XCDs, each
    // with its own L2.  The gx statistic blocks that read the same R tile get ids
    // congruent mod 8, i.e. the same XCD back to back: one of them misses in L2,
    // the others hit.
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int bx = slot % gx;
    const int yz = (slot / gx) * 8 + xcd;
    if (yz >= gy * gz) return;
    const int by = yz % gy, bz = yz / gy;
    const int tile0 = (bx * (kA16Threads / 64) + wave) * NQ;    // first statistic tile
    const int kc0 = by * (16 * MC);
    const int64_t tb = (int64_t)bz * frames_per_block;
    const int64_t te = tb + frames_per_block < nframes ? tb + frames_per_block : nframes;

    const int xs_elems = (D + 3) * kA16XS;                     // floats (+ a spare row)
    const int r_halves = 16 * MC * kA16RS;                 // per hi / lo image
    const size_t buf_bytes = (size_t)xs_elems * 4 + (size_t)r_halves * 2 * 2;
    auto xs_of = [&](int buf) { return reinterpret_cast<float*>(smem + buf * buf_bytes); };
    auto rh_of = [&](int buf) {
        return reinterpret_cast<_Float16*>(smem + buf * buf_bytes + (size_t)xs_elems * 4);
    };

    // the two X^T rows of this lane's statistic column in each of its tiles
    // and the power-of-two range scale of the product, sx[a] * sx[b]: applied when
    // the B fragment is generated (a scale looked up per staged element would be
    // an LDS read + wait in the middle of the MFMA stream)
    int ra[NQ], rb[NQ];
    float sab[NQ];
    auto factors = [&](int uu, int& a, int& b) {
        const int col = 16 * (tile0 + uu) + i, slab = col >> 2;
        a = b = Dp + 1;
        if (slab < nslab) {
            const int t = tab[slab];
            b = ((t >> 8) & 0xff) + (col & 3);
            a = (t >> 16) ? b : (t & 0xff);
        }
    };
#pragma unroll
    for (int uu = 0; uu < NQ; ++uu) {
        int a, b;
        factors(uu, a, b);
        // row D = ones, row D + 1 = zeros
        ra[uu] = (a < D ? a : (a == Dp ? D : D + 1)) * kA16XS;
        rb[uu] = (b < D ? b : (b == Dp ? D : D + 1)) * kA16XS;
        sab[uu] = (a < D ? sc[a] : 1.f) * (b < D ? sc[b] : 1.f);
    }
    f32x4 acc[MC][NQ];
#pragma unroll
    for (int c = 0; c < MC; ++c)
#pragma unroll
        for (int uu = 0; uu < NQ; ++uu) acc[c][uu] = f32x4{0, 0, 0, 0};

    // staging registers (global -> registers during the MFMAs -> other LDS buffer)
    constexpr int RPT = 16 * MC * (kA16FT / 2) / kA16Threads;    // (component, frame pair)
    constexpr int XPT = (kA16FT * 64 + kA16Threads - 1) / kA16Threads;   // D <= 64
    const int xcount = kA16FT * D;
    struct Stage { float x[XPT]; float r[RPT][2]; float w[HAS_SR ? RPT : 1][2]; };
    // staged item v of this thread: component kk (of 128) and frame pair fp (of
    // 32).  One wave instruction covers 32 components x 2 frame pairs: 128-byte
    // row segments from global memory, and ds_write_b32 with at most 2 lanes per
    // bank in the swizzled R^T image.
    auto stage_item = [&](int v, int& kk, int& fp) {
        const int slot = wave * RPT + v;
        constexpr int NB = MC / 2;                              // blocks of 32 components
        kk = 32 * (slot % NB) + (lane & 15) + 16 * (lane >> 5);
        fp = 2 * (slot / NB) + ((lane >> 4) & 1);
    };
    // Loads are unconditional on clamped addresses and nothing is computed on the
    // loaded values here: a branch or an early use per load makes hipcc put an
    // s_waitcnt behind every one of them (one memory round trip each).  Scaling
    // and zeroing of the out-of-range elements happen when the tile is stored.
    // Loads are unconditional on clamped addresses and nothing is computed on the
    // loaded values here: a branch or an early use per load makes hipcc put an
    // s_waitcnt behind every one of them (one memory round trip each).  Scaling
    // and zeroing of the out-of-range elements happen when the tile is stored.
    // Addresses are a uniform tile base + a 32-bit per-lane index.
    const int kvalid = K - kc0 < 16 * MC ? K - kc0 : 16 * MC;      // >= 1
    auto load_tile = [&](int64_t t0, Stage& st) {
        const float* xsrc = X + t0 * D;
        const int rows = (int)(te - t0 < kA16FT ? te - t0 : kA16FT);       // >= 1
        const int xlast = rows * D - 1;
#pragma unroll
        for (int v = 0; v < XPT; ++v) {
            const int idx = tid + v * kA16Threads;
            st.x[v] = xsrc[idx <= xlast ? idx : xlast];
        }
        const float* Rt = R + t0 * K + kc0;
        const float* Wt = HAS_SR ? SR + t0 * S : nullptr;
#pragma unroll
        for (int v = 0; v < RPT; ++v) {
            int kk, fp;
            stage_item(v, kk, fp);
            const int kcl = kk < kvalid ? kk : kvalid - 1;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int row = 2 * fp + h < rows ? 2 * fp + h : rows - 1;
                    st.r[v][h] = Rt[row * K + kcl];
                    if (HAS_SR) st.w[v][h] = Wt[row * S + (kc0 + kcl) / G];
                }
        }
        __builtin_amdgcn_sched_group_barrier(0x020, XPT + (HAS_SR ? 4 : 2) * RPT, 0);
    };
    auto store_x = [&](int buf, int64_t t0, const Stage& st, int v) {
        float* xs = xs_of(buf);
        const int xvalid = (int)(te - t0 < kA16FT ? te - t0 : kA16FT) * D;
        const int idx = tid + v * kA16Threads;
        const int f = idx / D, d = idx - f * D;
        // threads past the tile write to the spare row behind the constants
        const int at = idx < xcount ? d * kA16XS + f : (D + 2) * kA16XS + (tid & 63);
        xs[at] = idx < xvalid ? st.x[v] : 0.f;
    };
    auto store_r = [&](int buf, int64_t t0, const Stage& st, int v) {
        _Float16* rh = rh_of(buf);
        _Float16* rl = rh + r_halves;
        const int rows = (int)(te - t0 < kA16FT ? te - t0 : kA16FT);
        int kk, fp;
        stage_item(v, kk, fp);
        const int at = kk * kA16RS + (((fp >> 2) ^ (kk & 7)) << 3) + 2 * (fp & 3);
        float r2[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float val = st.r[v][h] * (float)(1 << kRespBits);
            if (HAS_SR) val *= st.w[v][h];
            r2[h] = (2 * fp + h < rows && kk < kvalid) ? val : 0.f;
        }
        hp2 hi, lo;
        split2(r2[0], r2[1], hi, lo);
        *reinterpret_cast<hp2*>(rh + at) = hi;
        *reinterpret_cast<hp2*>(rl + at) = lo;
    };
    auto store_tile = [&](int buf, int64_t t0, const Stage& st) {
#pragma unroll
        for (int v = 0; v < XPT; ++v) store_x(buf, t0, st, v);
#pragma unroll
        for (int v = 0; v < RPT; ++v) store_r(buf, t0, st, v);
    };

    for (int buf = 0; buf < 2; ++buf) {                       // constant rows, both buffers
        float* xs = xs_of(buf);
        for (int f = tid; f < kA16XS; f += kA16Threads) {
            xs[D * kA16XS + f] = 1.f;
            xs[(D + 1) * kA16XS + f] = 0.f;
        }
    }
    // Registers run two tiles ahead of the MFMAs, LDS one: while tile t is being
    // multiplied, tile t+1 sits in registers (stored to the other LDS buffer after
    // the MFMAs) and the loads of tile t+2 are in flight.
    const int64_t ntiles = (te - tb + kA16FT - 1) / kA16FT;
    Stage sa, sb;
    if (ntiles > 0) { load_tile(tb, sa); store_tile(0, tb, sa); }
    if (ntiles > 1) load_tile(tb + kA16FT, sa);
    __syncthreads();
    const bool active = tile0 * 16 < nq;         // waves past the last tile idle
    auto iteration = [&](int64_t tile, const Stage& held, Stage& far) {
        const int buf = (int)(tile & 1);
        const float* xs = xs_of(buf);
        const _Float16* rh = rh_of(buf);
        const _Float16* rl = rh + r_halves;
        // Unconditional (past the end the last tile is loaded again and never used):
        // behind a branch, hipcc merges the wait counters of the two paths and makes
        // every use of `held` wait for the loads just issued into `far` as well --
        // one full memory latency per tile.
        load_tile(tb + (tile + 2 < ntiles ? tile + 2 : ntiles - 1) * kA16FT, far);
        if (active) {
            // B fragments are produced one (k-step, statistic tile) ahead of the
            // MFMAs that consume them; sched_group_barrier spreads their LDS reads
            // and the ~28 VALU of the split into the issue gaps of the 24 MFMAs.
            auto gen_b = [&](int ks, int uu, h8& bh, h8& bl) {
                const int f0 = 32 * ks + 8 * g;                 // the lane's 8 frames
                const f32x4 xa0 = *reinterpret_cast<const f32x4*>(xs + ra[uu] + f0);
                const f32x4 xa1 = *reinterpret_cast<const f32x4*>(xs + ra[uu] + f0 + 4);
                const f32x4 xb0 = *reinterpret_cast<const f32x4*>(xs + rb[uu] + f0);
                const f32x4 xb1 = *reinterpret_cast<const f32x4*>(xs + rb[uu] + f0 + 4);
                split8((xa0 * sab[uu]) * xb0, (xa1 * sab[uu]) * xb1, bh, bl);
            };
            h8 bh[2], bl[2];
            gen_b(0, 0, bh[0], bl[0]);
            static_assert(NQ == 1 || NQ % 2 == 0, "the B double buffer alternates per tile");
#pragma unroll
            for (int ks = 0; ks < kA16FT / 32; ++ks) {
                h8 ah[MC], al[MC];
#pragma unroll
                for (int c = 0; c < MC; ++c) {
                    const int at = (16 * c + i) * kA16RS + (((4 * ks + g) ^ (i & 7)) << 3);
                    ah[c] = *reinterpret_cast<const h8*>(rh + at);
                    al[c] = *reinterpret_cast<const h8*>(rl + at);
                }
#pragma unroll
                for (int uu = 0; uu < NQ; ++uu) {
                    // (the generation after the last tile of the last k-step reads
                    // the next 32 frames of the padded rows and is never used)
                    const int cur = NQ == 1 ? 0 : (uu & 1);
                    const bool last = false;
                    if (NQ == 1) {
                        // no double buffer: generated in place after the MFMAs below
                    } else {
                        gen_b(uu + 1 < NQ ? ks : ks + 1, uu + 1 < NQ ? uu + 1 : 0, bh[cur ^ 1],
                              bl[cur ^ 1]);
                    }
#pragma unroll
                    for (int c = 0; c < MC; ++c)
                        acc[c][uu] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[c], bh[cur], acc[c][uu], 0, 0, 0);
#pragma unroll
                    for (int c = 0; c < MC; ++c)
                        acc[c][uu] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[c], bl[cur], acc[c][uu], 0, 0, 0);
#pragma unroll
                    for (int c = 0; c < MC; ++c)
                        acc[c][uu] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[c], bh[cur], acc[c][uu], 0, 0, 0);
                    if (NQ == 1) gen_b(ks + 1, 0, bh[0], bl[0]);
                    // the slice of the next tile's staging that belongs to this step
                    // (past the last tile this stores stale registers into the idle
                    // buffer: harmless, and it keeps the loop free of branches)
                    {
                        constexpr int NSTEP = (kA16FT / 32) * NQ;
                        const int step = ks * NQ + uu;
#pragma unroll
                        for (int v = step * RPT / NSTEP; v < (step + 1) * RPT / NSTEP; ++v)
                            store_r(buf ^ 1, tb + (tile + 1) * kA16FT, held, v);
#pragma unroll
                        for (int v = step * XPT / NSTEP; v < (step + 1) * XPT / NSTEP; ++v)
                            store_x(buf ^ 1, tb + (tile + 1) * kA16FT, held, v);
                    }
                    if (!last && NQ > 1) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);          // DS reads
                        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);          // MFMA
#pragma unroll
                        for (int z = 0; z < 16; ++z) {
                            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);      // VALU
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // MFMA
                        }
                    }
                }
                // keep the A fragments of the next k-step from being hoisted up here
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!active) store_tile(buf ^ 1, tb + (tile + 1) * kA16FT, held);
        __syncthreads();
    };
    for (int64_t tile = 0; tile < ntiles; tile += 2) {
        iteration(tile, sa, sb);
        if (tile + 1 < ntiles) iteration(tile + 1, sb, sa);
    }
    // C rows = components kc0 + 16 c + 4 g + r, columns = statistic 16 tile + i;
    // undo the frame scaling (one factor sx per real column) and the 2^12 of R
    const float* isx = sc + 64;
#pragma unroll
    for (int uu = 0; uu < NQ; ++uu) {
        const int q = (tile0 + uu) * 16 + i;
        if (q >= nq) continue;
        int a, b;
        factors(uu, a, b);
        const double unscale = (a < D ? (double)isx[a] : 1.0) * (b < D ? (double)isx[b] : 1.0) /
                               (double)(1 << kRespBits);
#pragma unroll
        for (int c = 0; c < MC; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = kc0 + 16 * c + 4 * g + r;
                if (k < K)
                    atomicAdd(Sp + (size_t)k * nq + q,
                              (double)acc[c][uu][r] * unscale);
            }
    }
}

// ---------------------------------------------------------------------------
// K2 for packed responsibilities (one mixture, no state factor).  Same tiles and
// products as acc16_kernel; what changes is the data movement, which bounds that
// kernel (a CU streams ~10 B/clk when every CU does: the 42 KB of a 64-frame
// tile take longer than the 768 MFMAs of 4 statistic tiles per wave):
//  * both operands arrive as ready-made LDS images -- R as K1 packed it
//    (estep_tiles.h: packed_word), X as xt_image_kernel transposed it -- so a
//    tile is a linear copy of 16-byte pieces, global_load_dwordx4 ->
//    ds_write_b128, without any address or select arithmetic;
//  * staging is a rolling window: every step of the MFMA loop stores the pieces
//    loaded half a tile earlier and issues the loads of the pieces half a tile
//    ahead (~half the registers of one tile instead of two whole tiles);
//  * the registers this frees allow two waves per SIMD (8 waves of 2 statistic
//    tiles): one wave's B-fragment arithmetic and LDS waits overlap the other's
//    MFMAs;
//  * A and B fragments are loaded one k-step / one step ahead, in place, and the
//    single barrier per tile sits in front of its last step (see `iteration`).
// ---------------------------------------------------------------------------
// X [T, D] -> per 64-frame tile the image [D + 2][kA16XS] of the range-scaled
// frames x_d * s_d (rows D, D + 1 = the constants 1, 0; frames past T = 0),
// padded to whole pieces
__global__ __launch_bounds__(256) void xt_image_kernel(int64_t nframes, int D, int NX,
                                                       const float* __restrict__ X,
                                                       const float* __restrict__ sc,
                                                       float* __restrict__ Xt) {
    __shared__ float tile[kA16FT * 65];
    __shared__ float scale[64];
    if (threadIdx.x < 64) scale[threadIdx.x] = sc[threadIdx.x];
    const int64_t tau = blockIdx.x, t0 = tau * kA16FT;
    const int rows = (int)(nframes - t0 < kA16FT ? nframes - t0 : kA16FT);
    for (int e = threadIdx.x; e < rows * D; e += 256) {
        const int f = e / D, d = e - f * D;
        tile[f * 65 + d] = X[t0 * D + e];
    }
    __syncthreads();
    float* out = Xt + tau * ((size_t)NX * (kPiece / 4));
    for (int e = threadIdx.x; e < NX * (kPiece / 4); e += 256) {
        const int row = e / kA16XS, col = e - row * kA16XS;
        float v = 0.f;
        if (row < D) v = col < rows ? tile[col * 65 + row] * scale[row] : 0.f;
        else if (row == D) v = 1.f;
        __builtin_nontemporal_store(v, out + e);
    }
}

template <int NQ, int NX, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void acc16p_kernel(
    int64_t nframes, int D, int K, int nslab, const float* __restrict__ Xt,
    const unsigned* __restrict__ Rimg, const int* __restrict__ tab,
    const float* __restrict__ sc, int64_t frames_per_block, double* __restrict__ Sp, int gx,
    int gy, int gz) {
    constexpr int MC = kA16MC;
    static_assert(16 * MC == kPackedComps && kA16FT == kPackedFrames && kA16RS == kPackedFrames,
                  "the packed image is this kernel's LDS tile");
    static_assert(kPackedRespBits == kRespBits, "K1 packs with the scale K2 removes");
    constexpr int NSTEP = (kA16FT / 32) * NQ;            // MFMA steps per tile
    constexpr int LAG = NSTEP / 2;                        // steps between load and store
    constexpr int NR = 16 * MC * kA16RS * 2 * 2 / kPiece;    // R pieces (hi + lo images)
    constexpr int NI = NX + NR;
    // 256 threads move one piece; with 8 waves the two halves of the workgroup take
    // alternate pieces (an odd last piece is moved twice: same bytes, same place)
    constexpr int HALVES = WAVES / 4, NIT = (NI + HALVES - 1) / HALVES;
    static_assert(WAVES == 4 || WAVES == 8, "one or two waves per SIMD");
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int D4 = d4_of(D), Dp = 4 * D4, nq = nslab * 4;
    // XCD-aware mapping as in acc16_kernel
    const int id = blockIdx.x, xcd = id & 7, slot0 = id >> 3;
    const int bx = slot0 % gx;
    const int yz = (slot0 / gx) * 8 + xcd;
    if (yz >= gy * gz) return;
    const int by = yz % gy, bz = yz / gy;
    const int tile0 = (bx * WAVES + wave) * NQ;
    const int kc0 = by * (16 * MC);
    const int64_t tb = (int64_t)bz * frames_per_block;
    const int64_t te = tb + frames_per_block < nframes ? tb + frames_per_block : nframes;
    const int ntiles = (int)((te - tb + kA16FT - 1) / kA16FT);
    const int64_t tau0 = tb / kA16FT, tau_last = (nframes + kA16FT - 1) / kA16FT - 1;

    // LDS: two buffers of [X image: NX pieces][R hi image][R lo image].  Every LDS
    // address below is one per-lane byte offset (kept in a register for the whole
    // kernel) plus a compile-time constant -- buffer, k-step, component tile, hi / lo
    // all go into the offset field of the DS instruction.
    constexpr int buf_bytes = NI * kPiece;
    constexpr int r_off = NX * kPiece, lo_off = 16 * MC * kA16RS * 2;

    auto factors = [&](int uu, int& a, int& b) {
        const int col = 16 * (tile0 + uu) + i, slab = col >> 2;
        a = b = Dp + 1;
        if (slab < nslab) {
            const int t = tab[slab];
            b = ((t >> 8) & 0xff) + (col & 3);
            a = (t >> 16) ? b : (t & 0xff);
        }
    };
    // byte offsets of the lane's 8 frames (k-step 0) in the two X^T rows of its
    // statistic column, per tile of the wave
    int xa_off[NQ], xb_off[NQ];
#pragma unroll
    for (int uu = 0; uu < NQ; ++uu) {
        int a, b;
        factors(uu, a, b);
        xa_off[uu] = ((a < D ? a : (a == Dp ? D : D + 1)) * kA16XS + 8 * g) * 4;
        xb_off[uu] = ((b < D ? b : (b == Dp ? D : D + 1)) * kA16XS + 8 * g) * 4;
    }
    // A fragments: row i of component tile c, chunk (4 ks + g) ^ (i & 7)
    int a_off[kA16FT / 32];
#pragma unroll
    for (int ks = 0; ks < kA16FT / 32; ++ks)
        a_off[ks] = r_off + (i * kA16RS + (((4 * ks + g) ^ (i & 7)) << 3)) * 2;
    const int t_off = 16 * (tid & (kA16Threads - 1));
    const int half = __builtin_amdgcn_readfirstlane(tid / kA16Threads);

    f32x4 acc[MC][NQ];
#pragma unroll
    for (int c = 0; c < MC; ++c)
#pragma unroll
        for (int uu = 0; uu < NQ; ++uu) acc[c][uu] = f32x4{0, 0, 0, 0};

    // piece v of a tile: v < NX from the X image, the others from the R image of
    // component block `by`; thread tid moves bytes [16 tid, 16 tid + 16) of it.
    // Tiles past the end of the data read the last tile again (never used).
    const int nblk = (K + 16 * MC - 1) / (16 * MC);
    // item j of a thread = piece HALVES * j + half
    u32x4 w[NIT];
    const char* xsrc = reinterpret_cast<const char*>(Xt) + t_off;
    const char* rsrc = reinterpret_cast<const char*>(Rimg) + t_off;
    auto piece_of = [&](int j) {
        const int v = HALVES * j + half;
        return v < NI ? v : NI - 1;
    };
    auto issue = [&](int tile, int j) {
        int64_t tau = tau0 + tile;
        tau = tau < tau_last ? tau : tau_last;
        const int v = piece_of(j);
        const char* src = v < NX ? xsrc + (tau * NX + v) * (size_t)kPiece
                                 : rsrc + ((tau * nblk + by) * NR + (v - NX)) * (size_t)kPiece;
        w[j] = *reinterpret_cast<const u32x4*>(src);
    };
    auto store = [&](int j, int buf) {
        *reinterpret_cast<u32x4*>(smem + t_off + (buf * buf_bytes + piece_of(j) * kPiece)) = w[j];
    };
    // step that stores item j: all but the last step of a tile, see the barrier below
    auto store_step = [](int j) { return NSTEP == 1 ? 0 : j * (NSTEP - 1) / NIT; };

    if (ntiles > 0) {
#pragma unroll
        for (int j = 0; j < NIT; ++j) issue(0, j);
#pragma unroll
        for (int j = 0; j < NIT; ++j) store(j, 0);
#pragma unroll
        for (int j = 0; j < NIT; ++j)
            if (store_step(j) < LAG) issue(1, j);
    }
    __syncthreads();
    const bool active = tile0 * 16 < nq;
    // operand fragments, loaded one step (B) / one k-step (A) before the MFMAs
    // that use them: loop-carried, the last step of a tile loads from the next one
    h8 ah[MC], al[MC], bh[2], bl[2];
    auto gen_b = [&](int buf, int ks, int uu, h8& h, h8& l) {
        const char* pa = smem + xa_off[uu] + (buf * buf_bytes + 128 * ks);
        const char* pb = smem + xb_off[uu] + (buf * buf_bytes + 128 * ks);
        const f32x4 xa0 = *reinterpret_cast<const f32x4*>(pa);
        const f32x4 xa1 = *reinterpret_cast<const f32x4*>(pa + 16);
        const f32x4 xb0 = *reinterpret_cast<const f32x4*>(pb);
        const f32x4 xb1 = *reinterpret_cast<const f32x4*>(pb + 16);
        split8(xa0 * xb0, xa1 * xb1, h, l);
    };
    auto a_ptr = [&](int buf, int ks, int c) {
        return smem + a_off[ks] + (buf * buf_bytes + c * 16 * kA16RS * 2);
    };
    if (active && ntiles > 0) {
#pragma unroll
        for (int c = 0; c < MC; ++c) {
            ah[c] = *reinterpret_cast<const h8*>(a_ptr(0, 0, c));
            al[c] = *reinterpret_cast<const h8*>(a_ptr(0, 0, c) + lo_off);
        }
        gen_b(0, 0, 0, bh[0], bl[0]);
    }
    // One tile out of buffer `buf` (a constant once inlined).  ONE barrier per
    // tile, in front of its last step: by then every wave has stored its pieces of
    // tile + 1 (all stores are in the steps before) and has read everything it
    // needs from this buffer (the fragments of the last step were loaded during
    // the step before), so the last step may load the first fragments of tile + 1
    // from the other buffer, and the next tile may overwrite this one.
    auto iteration = [&](int tile, int buf) __attribute__((always_inline)) {
        // step s: store the pieces of tile + 1 that are due, issue those due LAG
        // steps from now (of tile + 1, or of tile + 2 past the end of this tile)
        auto stage_step = [&](int s) {
#pragma unroll
            for (int j = 0; j < NIT; ++j)
                if (store_step(j) == s) store(j, buf ^ 1);
#pragma unroll
            for (int j = 0; j < NIT; ++j)
                if (store_step(j) == (s + LAG) % NSTEP) issue(tile + 1 + (s + LAG) / NSTEP, j);
        };
        if (active) {
#pragma unroll
            for (int ks = 0; ks < kA16FT / 32; ++ks) {
#pragma unroll
                for (int uu = 0; uu < NQ; ++uu) {
                    const int step = ks * NQ + uu;
                    const bool last_step = step == NSTEP - 1, last_uu = uu == NQ - 1;
                    if (last_step) __syncthreads();
                    // where the fragments of the next step / k-step come from
                    const int nbuf = last_step ? buf ^ 1 : buf;
                    const int nks = last_uu ? (ks + 1) % (kA16FT / 32) : ks;
                    const int cur = step & 1;
                    gen_b(nbuf, nks, last_uu ? 0 : uu + 1, bh[cur ^ 1], bl[cur ^ 1]);
#pragma unroll
                    for (int c = 0; c < MC; ++c)
                        acc[c][uu] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[c], bh[cur], acc[c][uu], 0, 0, 0);
#pragma unroll
                    for (int c = 0; c < MC; ++c) {
                        acc[c][uu] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[c], bl[cur], acc[c][uu], 0, 0, 0);
                        if (last_uu) ah[c] = *reinterpret_cast<const h8*>(a_ptr(nbuf, nks, c));
                    }
#pragma unroll
                    for (int c = 0; c < MC; ++c) {
                        acc[c][uu] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[c], bh[cur], acc[c][uu], 0, 0, 0);
                        if (last_uu)
                            al[c] = *reinterpret_cast<const h8*>(a_ptr(nbuf, nks, c) + lo_off);
                    }
                    stage_step(step);
                    // one step at a time: keeps the staging and the fragment loads
                    // of later steps from being hoisted (and their registers live)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < NSTEP; ++s) {
                if (s == NSTEP - 1) __syncthreads();
                stage_step(s);
            }
        }
    };
    for (int tile = 0; tile < ntiles; tile += 2) {
        iteration(tile, 0);
        if (tile + 1 < ntiles) iteration(tile + 1, 1);
    }
    const float* isx = sc + 64;
#pragma unroll
    for (int uu = 0; uu < NQ; ++uu) {
        const int q = (tile0 + uu) * 16 + i;
        if (q >= nq) continue;
        int a, b;
        factors(uu, a, b);
        const double unscale = (a < D ? (double)isx[a] : 1.0) * (b < D ? (double)isx[b] : 1.0) /
                               (double)(1 << kRespBits);
#pragma unroll
        for (int c = 0; c < MC; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = kc0 + 16 * c + 4 * g + r;
                if (k < K) atomicAdd(Sp + (size_t)k * nq + q, (double)acc[c][uu][r] * unscale);
            }
    }
}

// ---------------------------------------------------------------------------
// K2 for packed responsibilities, second form: LDS-DMA staging, 4 statistic tiles
// per wave.  acc16p_kernel above is bound by the CU's load path: every 64-frame
// tile (44 KB at D = 40) is fetched by the 4 workgroups that own the 4 blocks of
// statistic columns, through registers -- 5.7 GB per launch at config 2, ~10 B/clk
// per CU, which is what a CU streams when every CU does (MI355X_MICROARCH.md).
// Here
//  * a wave owns 4 statistic tiles (a workgroup 512 columns: 2 blocks instead of
//    4 cover the 928 columns of D = 40, so the R tiles cross L2 -> CU twice, not
//    four times), with the A fragments of the 8 component tiles loaded in two
//    halves so that 128 accumulators + 4 B fragments + 4 A fragments fit 256
//    registers at two waves per SIMD;
//  * the tiles are copied global -> LDS by the DMA path (global_load_lds_dwordx4:
//    the images are lane-linear by construction), no staging registers, no
//    ds_write; tile t + 1 is in flight while tile t is multiplied, one barrier per
//    tile.
// ---------------------------------------------------------------------------
// SR (mixture sets): the tiles hold the responsibilities WITHIN each state's mixture
// (what the E-step knows); the state posteriors of the forward-backward pass that ran
// in between arrive as transposed tiles Gt [64-frame tile][state][64 frames] and are
// multiplied in while the tile sits in LDS, each element once per workgroup:
// (hi + lo) * gamma in fp32, split again.  Three LDS buffers: tile t + 2 in flight
// (DMA), tile t + 1 being folded in place -- every thread its own 2 x 8 elements,
// spread over the steps before the barrier --, tile t multiplied.  lgG = log2 of the
// components per state (8 .. 128: the states of a 128-component block fit 4 KiB).
#ifndef BEER_K1_SCHED
#define BEER_K1_SCHED 0        // experiment builds (tools/ab_build.sh)
#endif
#ifndef BEER_SR_ABL
#define BEER_SR_ABL 0          // ablation builds only (tools/ab_build.sh)
#endif
template <int NX, bool SR>
__global__ __launch_bounds__(512, 2) void acc16d_kernel(
    int64_t nframes, int D, int K, int nslab, const float* __restrict__ Xt,
    const unsigned* __restrict__ Rimg, const int* __restrict__ tab,
    const float* __restrict__ sc, int64_t frames_per_block, double* __restrict__ Sp, int gx,
    int gy, int gz, const float* __restrict__ Gt, int lgG) {
    constexpr int MC = kA16MC, NQ = 4, WAVES = 8, NB = SR ? 3 : 2;
    constexpr int NR = 16 * MC * kA16RS * 2 * 2 / kPiece;        // R pieces (hi + lo images)
    constexpr int NI = NX + NR, NKB = NI * (kPiece / 1024);       // 1 KiB DMA blocks per tile
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int D4 = d4_of(D), Dp = 4 * D4, nq = nslab * 4;
    const int id = blockIdx.x, xcd = id & 7, slot0 = id >> 3;
    const int bx = slot0 % gx;
    const int yz = (slot0 / gx) * 8 + xcd;
    if (yz >= gy * gz) return;
    const int by = yz % gy, bz = yz / gy;
    const int tile0 = (bx * WAVES + wave) * NQ;
    const int kc0 = by * (16 * MC);
    const int64_t tb = (int64_t)bz * frames_per_block;
    const int64_t te = tb + frames_per_block < nframes ? tb + frames_per_block : nframes;
    const int ntiles = (int)((te - tb + kA16FT - 1) / kA16FT);
    const int64_t tau0 = tb / kA16FT;
    constexpr int buf_bytes = NI * kPiece;
    constexpr int r_off = NX * kPiece, lo_off = 16 * MC * kA16RS * 2;
    constexpr int g_base = NB * buf_bytes;                        // SR: NB x 4 KiB of gamma^T
    const int nblk = (K + 16 * MC - 1) / (16 * MC);
    const int gbytes = SR ? ((16 * MC) >> lgG) * kA16FT * 4 : 0;  // gamma^T of one block-tile
    const int gkb = (gbytes + 1023) >> 10;

    auto factors = [&](int uu, int& a, int& b) {
        const int col = 16 * (tile0 + uu) + i, slab = col >> 2;
        a = b = Dp + 1;
        if (slab < nslab) {
            const int t = tab[slab];
            b = ((t >> 8) & 0xff) + (col & 3);
            a = (t >> 16) ? b : (t & 0xff);
        }
    };
    int xa_off[NQ], xb_off[NQ];
#pragma unroll
    for (int uu = 0; uu < NQ; ++uu) {
        int a, b;
        factors(uu, a, b);
        xa_off[uu] = ((a < D ? a : (a == Dp ? D : D + 1)) * kA16XS + 8 * g) * 4;
        xb_off[uu] = ((b < D ? b : (b == Dp ? D : D + 1)) * kA16XS + 8 * g) * 4;
    }
    int a_off[kA16FT / 32];
#pragma unroll
    for (int ks = 0; ks < kA16FT / 32; ++ks)
        a_off[ks] = r_off + (i * kA16RS + (((4 * ks + g) ^ (i & 7)) << 3)) * 2;

    f32x4 acc[MC][NQ];
#pragma unroll
    for (int c = 0; c < MC; ++c)
#pragma unroll
        for (int uu = 0; uu < NQ; ++uu) acc[c][uu] = f32x4{0, 0, 0, 0};

    // DMA of tile `tile` into buffer `buf`: 1 KiB blocks kb = wave, wave + 8, ...;
    // blocks < 4 NX from the X image, the others from the R image of block `by`
    const char* xsrc = reinterpret_cast<const char*>(Xt);
    const char* rsrc = reinterpret_cast<const char*>(Rimg);
    auto stage = [&](int tile, int buf) {
        const int64_t tau = tau0 + tile;
#pragma unroll
        for (int n = 0; n < (NKB + (SR ? 4 : 0) + WAVES - 1) / WAVES; ++n) {
            const int kb = wave + WAVES * n;
            if (kb < NKB) {
                const char* src = kb < 4 * NX
                    ? xsrc + (tau * NX) * (size_t)kPiece + (size_t)kb * 1024
                    : rsrc + ((tau * nblk + by) * NR) * (size_t)kPiece + (size_t)(kb - 4 * NX) * 1024;
                __builtin_amdgcn_global_load_lds(
                    reinterpret_cast<const u32x4*>(src) + lane,
                    (__attribute__((address_space(3))) void*)(smem + buf * buf_bytes + kb * 1024),
                    16, 0, 0);
            } else if (SR && kb - NKB < gkb) {
                const char* src = reinterpret_cast<const char*>(Gt) +
                                  (tau * nblk + by) * (size_t)gbytes + (size_t)(kb - NKB) * 1024;
                __builtin_amdgcn_global_load_lds(
                    reinterpret_cast<const u32x4*>(src) + lane,
                    (__attribute__((address_space(3))) void*)(smem + g_base + buf * 4096 +
                                                              (kb - NKB) * 1024),
                    16, 0, 0);
            }
        }
    };
    // SR: quarter q (of 4) of this thread's share of the R image in `buf`: 4 frames of
    // row c (chunk position pos = frames 8 (pos ^ (c & 7)) .., half q & 1 of it), hi and
    // lo, times the state's gamma.  A quarter at a time: few registers live at once.
    auto fold = [&](int buf, int q) {
#if BEER_SR_ABL == 1
        return;
#endif
        int tq = tid;                       // (opaque: the addresses are recomputed per
        asm volatile("" : "+v"(tq));        //  call, not kept in registers across the tile)
        const int p = tq + 512 * (q >> 1), c = p >> 3, pos = p & 7, ch = pos ^ (c & 7);
        char* ph = smem + buf * buf_bytes + r_off + c * (kA16RS * 2) + pos * 16 + 8 * (q & 1);
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        const h4 hi = *reinterpret_cast<const h4*>(ph);
        const h4 lo = *reinterpret_cast<const h4*>(ph + lo_off);
        const f32x4 gm = *reinterpret_cast<const f32x4*>(
            reinterpret_cast<const float*>(smem + g_base + buf * 4096) +
            ((c >> lgG) * kA16FT + 8 * ch + 4 * (q & 1)));
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaf((float)hi[e], gm[e], (float)lo[e] * gm[e]);
        hp2 h0, l0, h1, l1;
        split2(v[0], v[1], h0, l0);
        split2(v[2], v[3], h1, l1);
#if BEER_SR_ABL == 2
        if (gm[0] == 12345.f)
#endif
        {
            *reinterpret_cast<h4*>(ph) = h4{h0[0], h0[1], h1[0], h1[1]};
            *reinterpret_cast<h4*>(ph + lo_off) = h4{l0[0], l0[1], l1[0], l1[1]};
        }
    };
    if (ntiles > 0) stage(0, 0);
    if (SR && ntiles > 1) stage(1, 1);
    __syncthreads();
    if (SR && ntiles > 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) fold(0, q);
        __syncthreads();
    }
    const bool active = tile0 * 16 < nq;
    // Operand fragments are loaded IN PLACE one step ahead of the MFMAs that use them:
    // the A fragments of the next half of the component tiles during the last
    // statistic tile of the current half (each register right after its last use),
    // the B fragment of statistic tile uu for the next k-step as soon as the second
    // half is done with it.  No second register set, no exposed LDS latency.
    h8 ah[MC / 2], al[MC / 2], bh[NQ], bl[NQ];
    auto a_ptr = [&](int b, int ks, int c) {
        return smem + a_off[ks] + (b * buf_bytes + c * 16 * kA16RS * 2);
    };
    auto gen_b = [&](int b, int ks, int uu, h8& h, h8& l) {
        const char* pa = smem + xa_off[uu] + (b * buf_bytes + 128 * ks);
        const char* pb = smem + xb_off[uu] + (b * buf_bytes + 128 * ks);
        const f32x4 xa0 = *reinterpret_cast<const f32x4*>(pa);
        const f32x4 xa1 = *reinterpret_cast<const f32x4*>(pa + 16);
        const f32x4 xb0 = *reinterpret_cast<const f32x4*>(pb);
        const f32x4 xb1 = *reinterpret_cast<const f32x4*>(pb + 16);
        split8(xa0 * xb0, xa1 * xb1, h, l);
    };
    if (active && ntiles > 0) {
#pragma unroll
        for (int c = 0; c < MC / 2; ++c) {
            ah[c] = *reinterpret_cast<const h8*>(a_ptr(0, 0, c));
            al[c] = *reinterpret_cast<const h8*>(a_ptr(0, 0, c) + lo_off);
        }
#pragma unroll
        for (int uu = 0; uu < NQ; ++uu) gen_b(0, 0, uu, bh[uu], bl[uu]);
    }
    // One tile out of buffer `buf` (a constant once inlined): 16 steps of 12 MFMAs.
    // ONE barrier per tile, in front of step 12: by then this wave has read
    // everything it needs from `buf` (the operands of steps 12..15 were loaded
    // before), so the steps behind it may load the first operands of tile + 1 from
    // the other buffer -- whose DMA, issued at the start of this tile, the barrier's
    // vmcnt(0) has seen land -- and the next tile may overwrite `buf`.
    auto iteration = [&](int tile, int buf) __attribute__((always_inline)) {
        const int next = (buf + 1) % NB;
        if (SR) { if (tile + 2 < ntiles) stage(tile + 2, (buf + 2) % NB); }
        else if (tile + 1 < ntiles) stage(tile + 1, next);
        const bool fold_next = SR && tile + 1 < ntiles;
        if (active) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int half = 0; half < 2; ++half)
#pragma unroll
                    for (int uu = 0; uu < NQ; ++uu) {
                        const int st = (ks * 2 + half) * NQ + uu;
                        if (st == 3 * NQ) __syncthreads();
                        if (SR && st >= 1 && st <= 10 && st % 3 == 1 && fold_next)
                            fold(next, st / 3);
                        const bool last_uu = uu == NQ - 1;
                        // where the next A half / the next B fragment come from
                        const int nks = half ? (ks + 1) & 1 : ks, nhalf = half ^ 1;
                        const int nbuf = (half && ks == 1) ? next : buf;
#pragma unroll
                        for (int c = 0; c < MC / 2; ++c) {
                            f32x4& d = acc[half * (MC / 2) + c][uu];
                            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[c], bh[uu], d, 0, 0, 0);
                        }
#pragma unroll
                        for (int c = 0; c < MC / 2; ++c) {
                            f32x4& d = acc[half * (MC / 2) + c][uu];
                            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[c], bl[uu], d, 0, 0, 0);
                            if (last_uu)
                                ah[c] = *reinterpret_cast<const h8*>(
                                    a_ptr(nbuf, nks, nhalf * (MC / 2) + c));
                        }
#pragma unroll
                        for (int c = 0; c < MC / 2; ++c) {
                            f32x4& d = acc[half * (MC / 2) + c][uu];
                            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[c], bh[uu], d, 0, 0, 0);
                            if (last_uu)
                                al[c] = *reinterpret_cast<const h8*>(
                                    a_ptr(nbuf, nks, nhalf * (MC / 2) + c) + lo_off);
                        }
                        if (half) gen_b(nbuf, nks, uu, bh[uu], bl[uu]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
        } else {
            if (fold_next) {
#pragma unroll
                for (int q = 0; q < 4; ++q) fold(next, q);
            }
            __syncthreads();
        }
    };
    for (int tile = 0; tile < ntiles; tile += NB) {
        iteration(tile, 0);
        if (tile + 1 < ntiles) iteration(tile + 1, 1);
        if (NB == 3 && tile + 2 < ntiles) iteration(tile + 2, 2);
    }
    const float* isx = sc + 64;
#pragma unroll
    for (int uu = 0; uu < NQ; ++uu) {
        const int q = (tile0 + uu) * 16 + i;
        if (q >= nq) continue;
        int a, b;
        factors(uu, a, b);
        const double unscale = (a < D ? (double)isx[a] : 1.0) * (b < D ? (double)isx[b] : 1.0) /
                               (double)(1 << kRespBits);
#pragma unroll
        for (int c = 0; c < MC; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = kc0 + 16 * c + 4 * g + r;
                if (k < K) atomicAdd(Sp + (size_t)k * nq + q, (double)acc[c][uu][r] * unscale);
            }
    }
}

// State posteriors [T, S] -> transposed tiles Gt [tile of 64 frames][Spad states][64
// frames] (Spad = states of the padded component blocks; frames >= T and states >= S
// are 0): what acc16d_kernel<.., SR> copies to LDS next to a tile of responsibilities.
__global__ __launch_bounds__(256) void gt_image_kernel(int64_t nframes, int S, int Spad,
                                                       const float* __restrict__ sr,
                                                       float* __restrict__ Gt) {
    __shared__ float tile[64 * 65];
    const int64_t tau = blockIdx.x, t0 = tau * kA16FT;
    const int s0 = blockIdx.y * 64;
    for (int idx = threadIdx.x; idx < 64 * 64; idx += 256) {
        const int f = idx >> 6, st = idx & 63;
        tile[f * 65 + st] = (t0 + f < nframes && s0 + st < S) ? sr[(t0 + f) * S + s0 + st] : 0.f;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 64 * 64; idx += 256) {
        const int st = idx >> 6, f = idx & 63;
        if (s0 + st < Spad) Gt[(tau * Spad + s0 + st) * kA16FT + f] = tile[f * 65 + st];
    }
}

// packed responsibilities -> float32 (tests, callers that want to look at them)
__global__ void unpack_resps_kernel(int64_t nframes, int K, const unsigned* __restrict__ Rimg,
                                    float* __restrict__ R) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // (frame, k)
    if (idx >= nframes * K) return;
    const int64_t f = idx / K;
    const int k = (int)(idx - f * K);
    const int nblk = (K + kPackedComps - 1) / kPackedComps;
    const int f6 = (int)(f % kPackedFrames);
    const _Float16* hi = reinterpret_cast<const _Float16*>(
        Rimg + kPackedHeader / 4 + packed_word(f / kPackedFrames, nblk, k / kPackedComps, k % kPackedComps, f6 & ~1));
    const _Float16* lo = hi + kPackedComps * kPackedFrames;
    R[idx] = ((float)hi[f6 & 1] + (float)lo[f6 & 1]) * (1.f / (float)(1 << kRespBits));
}

// float32 responsibilities [T, K] (times state responsibilities [T, S]) -> packed
// tiles: one workgroup per tile of 64 frames x 128 components, transposed through
// LDS (rows of 129 words: conflict-free both ways), split, stored as whole
// 16-byte chunks.  For accumulations that are far from the memory roofline with
// float32 operands (full covariance: 4 statistic blocks re-read R) the 2 x 4 B
// per element of this pass buy the packed kernel.
__global__ __launch_bounds__(256) void pack_resps_kernel(int64_t nframes, int K, int S, int G,
                                                         const float* __restrict__ R,
                                                         const float* __restrict__ SR,
                                                         unsigned* __restrict__ out) {
    __shared__ float tile[kPackedFrames * 129];
    const int64_t tau = blockIdx.x, t0 = tau * kPackedFrames;
    const int beta = blockIdx.y, nblk = gridDim.y, kc0 = beta * kPackedComps;
    const int rows = (int)(nframes - t0 < kPackedFrames ? nframes - t0 : kPackedFrames);
    const float up = (float)(1 << kRespBits);
    for (int e = threadIdx.x; e < kPackedFrames * (kPackedComps / 4); e += 256) {
        const int f = e / (kPackedComps / 4), c4 = 4 * (e - f * (kPackedComps / 4)), k = kc0 + c4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (f < rows && k < K) {                               // K % 4 == 0
            v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(R + (t0 + f) * K + k)) * up;
            if (SR) {
                const float* sr = SR + (t0 + f) * S;
                if ((G & 3) == 0) {
                    v *= sr[k / G];
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] *= sr[(k + j) / G];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) tile[f * 129 + c4 + j] = v[j];
    }
    __syncthreads();
    typedef unsigned uint4_t __attribute__((ext_vector_type(4)));
    for (int e = threadIdx.x; e < kPackedComps * 8; e += 256) {
        const int kk = e >> 3, c8 = e & 7;
        f32x4 p0, p1;
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2) {
            p0[i2] = tile[(8 * c8 + i2) * 129 + kk];
            p1[i2] = tile[(8 * c8 + 4 + i2) * 129 + kk];
        }
        h8 hi, lo;
        split8(p0, p1, hi, lo);
        unsigned* dst = out + packed_word(tau, nblk, beta, kk, 8 * c8);
        __builtin_nontemporal_store(__builtin_bit_cast(uint4_t, hi), reinterpret_cast<uint4_t*>(dst));
        __builtin_nontemporal_store(
            __builtin_bit_cast(uint4_t, lo),
            reinterpret_cast<uint4_t*>(dst + kPackedComps * kPackedFrames / 2));
    }
}

// ---------------------------------------------------------------------------
// Fused accumulation for mixture sets with few statistics per Gaussian (diagonal
// / isotropic covariances): S[k, q] += sum_t r[t,k] sr[t, k / G] PHI_q(x_t) WITHOUT
// the responsibilities in memory.  The HMM iteration used to write the component
// responsibilities R [T, K] in its emission E-step and read them back here:
// 15.4 GB per 1 M frames at K = 1920 against 160 B / frame of input (VERDICT r1:
// 96x the algorithmic traffic).  Now the E-step only leaves the per-state
// log-normalisers [T, S]; after the forward-backward pass this kernel recomputes
// the component logits of a tile of 32 frames x 16 NTC components on the matrix
// cores (the k-loop of llh16_kernel), turns them into r sr 2^12 =
// exp(l - log_norm[t, s]) sr[t, s] 2^12 in registers -- no maximum, no sum: the
// normaliser is known -- and feeds them straight back to the matrix cores as the
// A operand of the statistics product: the C layout of the logits (lane (i, g):
// component i, frames 4g..4g+3 of both 16-frame tiles) IS the A layout of a
// 16 x 32 [component x frame] operand when the 32 frames of the contraction are
// taken in the order (tile 0: 4g..4g+3, tile 1: 4g..4g+3), and the B operand
// PHI_q(x_f) is generated from the transposed frame tile in that same order.
// A wave keeps its 16 NTC x 16 NQT statistics tile in registers over all the
// frame tiles it walks (fp32, <= 4096 frames), then adds it to the fp64 image.
// Bound: MFMA (192 + 144 per tile at D = 40 diagonal) + ~700 VALU around them;
// HBM: X (re-read per component chunk from L2), log_norm and sr once.
// ---------------------------------------------------------------------------
constexpr int kAfXS = 36;                 // row stride (floats) of the transposed frame tile
constexpr int kAfMaxFramesPerWave = 4096; // fp32 roundings per sum: 128

template <int NTC, int NQT, bool G4, int WAVES, int kXP>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void accf_kernel(
    int64_t nframes, int D, int K, int S, int G, int Greal, int nk, int nslab,
    const float* __restrict__ X, const _Float16* __restrict__ Pall,
    const float* __restrict__ inv_scale, const float* __restrict__ sc,
    const int* __restrict__ tab, const float* __restrict__ log_norm,
    const float* __restrict__ sr, int64_t frames_per_block, double* __restrict__ Sp, int dbg) {
    constexpr int MT = 2, FW = 32, QT = NTC / 4, NTHREADS = 64 * WAVES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int64_t bx;
    int by;
    {
        const int nch = (K + 16 * NTC - 1) / (16 * NTC);      // (24 KB of parameters each)
        if (!xcd_block((nframes + frames_per_block - 1) / frames_per_block, nch, nch, bx, by))
            return;
    }
    const int D4 = d4_of(D), Dp = 4 * D4, LD = ld16_of(D), nq = nslab * 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int xt_floats = (D + 2) * kAfXS;
    // LDS: the chunk's packed parameters (nk x NTC blocks of 2 KiB, read by every
    // wave for every frame tile: from L2 their latency sat in front of each handful
    // of MFMAs), the slab table, then per wave the frame tile row-major (A
    // fragments of the logits) and transposed (B fragments of the statistics)
    const int p_u4 = nk * NTC * 128;
    u4* Ps = reinterpret_cast<u4*>(smem);
    int* tabs = reinterpret_cast<int*>(Ps + p_u4);
    float* scales = reinterpret_cast<float*>(tabs + (nk + 1) * 8);      // [64] frame scales
    float* xw = scales + 64 + wave * (FW * LD + xt_floats);
    float* xt = xw + FW * LD;
    if (tid < 64) scales[tid] = sc[tid];
    {
        const u4* src = reinterpret_cast<const u4*>(Pall + (size_t)by * nk * NTC * 1024);
        for (int idx = tid; idx < p_u4; idx += NTHREADS) Ps[idx] = src[idx];
    }
    for (int idx = tid; idx < (nk + 1) * 8; idx += NTHREADS) tabs[idx] = tab[idx];
    {   // constant rows / columns, once
        for (int r = lane; r < FW; r += 64)
#pragma unroll 1
            for (int c = D; c < LD; ++c)
                xw[r * LD + c] = c == Dp ? 1.f : (c == Dp + 1 ? kConstEps : 0.f);
        for (int f = lane; f < kAfXS; f += 64) {
            xt[D * kAfXS + f] = 1.f;
            xt[(D + 1) * kAfXS + f] = 0.f;
        }
    }
    __syncthreads();

    const int kbase = by * (16 * NTC);
    const int nk_used = (nslab + 7) / 8;
    const int64_t tb = bx * frames_per_block;
    const int64_t te = tb + frames_per_block < nframes ? tb + frames_per_block : nframes;
    const u4* Pl = Ps + lane;
    const int* tl = tabs + 2 * g;
    const float* xrow[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) xrow[m] = xw + (m * 16 + i) * LD;

    // per lane: the inverse packing scale of its columns, statistic rows of its B columns
    float c1[NTC];
#pragma unroll
    for (int q = 0; q < QT; ++q) {
        const f32x4 inv = *reinterpret_cast<const f32x4*>(inv_scale + kbase + 64 * q + 4 * i);
#pragma unroll
        for (int j = 0; j < 4; ++j) c1[4 * q + j] = inv[j];
    }
    auto factors = [&](int uu, int& a, int& b) {
        const int col = 16 * uu + i, slab = col >> 2;
        a = b = Dp + 1;
        if (slab < nslab) {
            const int t = tabs[slab];
            b = ((t >> 8) & 0xff) + (col & 3);
            a = (t >> 16) ? b : (t & 0xff);
        }
    };
    int xa_off[NQT], xb_off[NQT];
#pragma unroll
    for (int uu = 0; uu < NQT; ++uu) {
        int a, b;
        factors(uu, a, b);
        xa_off[uu] = (a < D ? a : (a == Dp ? D : D + 1)) * kAfXS + 4 * g;
        xb_off[uu] = (b < D ? b : (b == Dp ? D : D + 1)) * kAfXS + 4 * g;
    }
    // states of the lane's components (G4: one per block of 64 components)
    constexpr int NST = G4 ? 1 : 4;
    int st_of[QT][NST];
#pragma unroll
    for (int q = 0; q < QT; ++q)
#pragma unroll
        for (int j = 0; j < NST; ++j) {
            const int s0 = (kbase + 64 * q + 4 * i + j) / G;
            st_of[q][j] = s0 < S ? s0 : S - 1;
        }

    f32x4 sacc[NTC][NQT];
#pragma unroll
    for (int c = 0; c < NTC; ++c)
#pragma unroll
        for (int uu = 0; uu < NQT; ++uu) sacc[c][uu] = f32x4{0, 0, 0, 0};

    struct AFrag { unsigned hi[MT][4], lo[MT][4]; };
    auto frag = [](const unsigned (&w)[4]) {
        return __builtin_bit_cast(h8, u4{w[0], w[1], w[2], w[3]});
    };
    auto make_half = [&](int s, int m, int h, AFrag& f) {
        const int t = tl[8 * s + h];
        const int a = t & 0xff, j = (t >> 8) & 0xff;
        const bool sq = (t >> 16) != 0;
        const f32x4 bb = *reinterpret_cast<const f32x4*>(xrow[m] + j);
        const float xx = xrow[m][a];
        f32x4 p;
#pragma unroll
        for (int e = 0; e < 4; ++e) p[e] = bb[e] * (sq ? bb[e] : xx);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            hp2 hh, ll;
            split2(p[2 * e], p[2 * e + 1], hh, ll);
            f.hi[m][2 * h + e] = __builtin_bit_cast(unsigned, hh);
            f.lo[m][2 * h + e] = __builtin_bit_cast(unsigned, ll);
        }
    };

    const int C4 = D >> 2;
    const bool rows4 = (D & 3) == 0;
    const f32x4* X4 = reinterpret_cast<const f32x4*>(X);
    // The tile's 32 x D floats are contiguous: 16-byte piece `lane + 64 it` of it goes
    // to row r, columns 4 c4 .. of the row-major image and to rows 4 c4 .., column r of
    // the transposed one.  The piece -> (r, c4) map does not depend on the tile: its
    // LDS offsets and scales are computed once.  All kXP loads of a tile are issued
    // back to back on clamped addresses (a loop that loads, waits and stores a piece
    // at a time put one memory latency per piece into every tile: 5 us of its 6.5).
    // (kXP pieces per lane: 5 for D <= 40, 8 for D <= 64)
    int xrc[kXP];                                            // r | c4 << 8, r = FW: no piece
    const int npieces = FW * C4;
#pragma unroll
    for (int it = 0; it < kXP; ++it) {
        const int idx = lane + 64 * it, r = idx / (C4 > 0 ? C4 : 1), c4 = idx - r * C4;
        xrc[it] = (rows4 && idx < npieces) ? (r | (c4 << 8)) : FW;
    }
    const float* scl = scales;
    // Second mapping, when a row is an even number of pieces (D = 8, 16, 24, 32, 40 ..):
    // lane = (row r, half h of the row), pieces h, h + 2, h + 4 .. of it.  Row and half
    // are fixed per lane, so every address below is ONE per-lane base plus a
    // compile-time constant per piece -- global source, scales, row-major and
    // transposed destinations -- where the linear mapping above spends ~30 VALU per
    // piece on decoding (r, c4) and forming five addresses (a third of the kernel's
    // VALU instructions went into this staging).  Rows past the end are read from the
    // last row: their responsibilities get weight 0 below, any finite value will do.
    const bool rowlane = rows4 && (C4 & 1) == 0 && (C4 >> 1) <= kXP;
    const int NP = C4 >> 1, lr = lane & 31, lh = lane >> 5;
    float* xw_l = xw + lr * LD + 4 * lh;                       // + 8 it
    float* xt_l = xt + 4 * lh * kAfXS + lr;                    // + (8 it + j) kAfXS
    const float* scl_l = scl + 4 * lh;                         // + 8 it
    // frame tiles of this wave: tb + 32 (wave + WAVES n)
    for (int64_t fb = tb + (int64_t)wave * FW; fb < te; fb += WAVES * FW) {
        const int rows = (int)(te - fb < FW ? te - fb : FW);              // >= 1
        // ---- the scaled frame tile, row-major and transposed (wave-private LDS) ----
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        f32x4 xv[kXP];
        if (rowlane) {
            const f32x4* src = X4 + (fb + (lr < rows ? lr : rows - 1)) * C4 + lh;
#pragma unroll
            for (int it = 0; it < kXP; ++it) xv[it] = src[2 * (it < NP ? it : NP - 1)];
        } else if (rows4) {
            const f32x4* Xt4 = X4 + fb * C4;
            const int last = rows * C4 - 1;
#pragma unroll
            for (int it = 0; it < kXP; ++it) {
                const int idx = lane + 64 * it;
                xv[it] = Xt4[idx < last ? idx : last];
            }
        } else {
            const float* Xt = X + fb * D;
            for (int idx = lane; idx < FW * D; idx += 64) {
                const int r = idx / D, c = idx - r * D;
                const float v = r < rows ? Xt[idx] * sc[c] : 0.f;
                xw[r * LD + c] = v;
                xt[c * kAfXS + r] = v;
            }
        }
        // the per-state normalisers and posteriors of the tile's rows, loaded now and
        // used after the k-loop: (frame 16 m + 4 g + r, state of the lane's components)
        // (unconditional loads on clamped rows, validity applied to constants: a select
        // or a branch on the loaded value makes hipcc load and wait one at a time)
        float nl2[QT][MT][4][NST], wg[QT][MT][4][NST];
        {
            const float* ln_t = log_norm + fb * S;
            const float* sr_t = sr + fb * S;
#pragma unroll
            for (int q = 0; q < QT; ++q)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * m + 4 * g + r;
                        const int rc = row < rows ? row : rows - 1;
#pragma unroll
                        for (int j = 0; j < NST; ++j) nl2[q][m][r][j] = ln_t[rc * S + st_of[q][j]];
                    }
            if (sr) {
#pragma unroll
                for (int q = 0; q < QT; ++q)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = 16 * m + 4 * g + r;
                            const int rc = row < rows ? row : rows - 1;
#pragma unroll
                            for (int j = 0; j < NST; ++j) wg[q][m][r][j] = sr_t[rc * S + st_of[q][j]];
                        }
            } else {
#pragma unroll
                for (int q = 0; q < QT; ++q)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int j = 0; j < NST; ++j) wg[q][m][r][j] = 1.f;
            }
        }
        // A tile none of whose frames gives the chunk's states any posterior contributes
        // exactly nothing: skip it (alignment graphs: most of the model's states are
        // absent from an utterance, their posteriors are exact zeros)
        if (sr) {
            float any = 0.f;
#pragma unroll
            for (int q = 0; q < QT; ++q)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int j = 0; j < NST; ++j)
                            any = __builtin_fmaxf(any, __builtin_fabsf(wg[q][m][r][j]));
#ifndef BEER_ACCF_NOSKIP
            if (__builtin_amdgcn_ballot_w64(any != 0.f) == 0) continue;
#endif
        }
        if (rowlane) {
#pragma unroll
            for (int it = 0; it < kXP; ++it) {
                if (it >= NP) break;
                const f32x4 v = xv[it] * *reinterpret_cast<const f32x4*>(scl_l + 8 * it);
                *reinterpret_cast<f32x4*>(xw_l + 8 * it) = v;
#pragma unroll
                for (int j = 0; j < 4; ++j) xt_l[(8 * it + j) * kAfXS] = v[j];
            }
        } else if (rows4) {
#pragma unroll
            for (int it = 0; it < kXP; ++it) {
                int rc = xrc[it];
                // (opaque: keeps the 5 LDS addresses per piece from being hoisted out of
                // the tile loop and held in 40 registers)
                asm volatile("" : "+v"(rc));
                const int r = rc & 0xff, c4 = rc >> 8;
                if (r >= FW) continue;                          // no such piece (tile-invariant)
                const f32x4 sv = *reinterpret_cast<const f32x4*>(scl + 4 * c4);
                const f32x4 v = r < rows ? xv[it] * sv : f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(xw + r * LD + 4 * c4) = v;
#pragma unroll
                for (int j = 0; j < 4; ++j) xt[(4 * c4 + j) * kAfXS + r] = v[j];
            }
        }
        // rows past the end: weight 0, and a normaliser that keeps the exponential at 0
        // (0 x inf would be NaN)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = 16 * m + 4 * g + r < rows;
                const float pen = ok ? 0.f : -1.0e30f, mult = ok ? (float)(1 << kRespBits) : 0.f;
#pragma unroll
                for (int q = 0; q < QT; ++q)
#pragma unroll
                    for (int j = 0; j < NST; ++j) {
                        nl2[q][m][r][j] = pen - nl2[q][m][r][j];
                        wg[q][m][r][j] *= mult;
                    }
            }
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");

        // ---- logits of 32 frames x 16 NTC components ----
        f32x4 acc[MT][NTC];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int c = 0; c < NTC; ++c) acc[m][c] = f32x4{0, 0, 0, 0};
        AFrag f0, f1;
        auto kstep = [&](int s, const AFrag& cur, AFrag& nxt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // a slice of the next k-step's A fragments (the table is padded by one k-step)
#pragma unroll
                for (int hh = q * MT * 2 / 4; hh < (q + 1) * MT * 2 / 4; ++hh)
                    make_half(s + 1, hh % MT, hh / MT, nxt);
#pragma unroll
                for (int c = q * QT; c < (q + 1) * QT; ++c) {
                    const h8 bhi = __builtin_bit_cast(h8, Pl[(s * NTC + c) * 128]);
                    const h8 blo = __builtin_bit_cast(h8, Pl[(s * NTC + c) * 128 + 64]);
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        acc[m][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(frag(cur.hi[m]), bhi, acc[m][c], 0, 0, 0);
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        acc[m][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(frag(cur.hi[m]), blo, acc[m][c], 0, 0, 0);
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        acc[m][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(frag(cur.lo[m]), bhi, acc[m][c], 0, 0, 0);
                }
            }
        };
#pragma unroll
        for (int hh = 0; hh < MT * 2; ++hh) make_half(0, hh % MT, hh / MT, f0);
        // (only the k-steps that hold slabs: the image's padding to an even count is zeros)
        if (!(dbg & 1))
        for (int s = 0; s < nk_used; s += 2) {
            kstep(s, f0, f1);
            if (s + 1 < nk_used) kstep(s + 1, f1, f0);
        }

        // ---- r sr 2^12 = exp(l - log_norm) sr 2^12, split into the A fragments ----
        // A fragment of component tile nt: words 0, 1 = frames 4g..4g+3 of tile 0,
        // words 2, 3 = the same rows of tile 1
        unsigned ah[NTC][4], al[NTC][4];
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt) {
            const int q = nt >> 2, jj = G4 ? 0 : (nt & 3);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    // l - log_norm first (one rounding of a small difference), THEN the change
                    // of base: scaling l (|l| ~ 100) and log_norm separately by a rounded
                    // log2(e) left a systematic 4e-6 in r
                    v[r] = __builtin_amdgcn_exp2f(
                               __builtin_fmaf(acc[m][nt][r], c1[nt], nl2[q][m][r][jj]) *
                               1.44269504088896340736f) *
                           wg[q][m][r][jj];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    hp2 hh, ll;
                    split2(v[2 * e], v[2 * e + 1], hh, ll);
                    ah[nt][2 * m + e] = __builtin_bit_cast(unsigned, hh);
                    al[nt][2 * m + e] = __builtin_bit_cast(unsigned, ll);
                }
            }
        }

        // ---- statistics: sacc[c][uu] += A'(c) x B'(uu) ----
        auto gen_b = [&](int uu, h8& bh, h8& bl) {
            const f32x4 xa0 = *reinterpret_cast<const f32x4*>(xt + xa_off[uu]);
            const f32x4 xa1 = *reinterpret_cast<const f32x4*>(xt + xa_off[uu] + 16);
            const f32x4 xb0 = *reinterpret_cast<const f32x4*>(xt + xb_off[uu]);
            const f32x4 xb1 = *reinterpret_cast<const f32x4*>(xt + xb_off[uu] + 16);
            split8(xa0 * xb0, xa1 * xb1, bh, bl);
        };
        h8 bh[2], bl[2];
        gen_b(0, bh[0], bl[0]);
        if (!(dbg & 2)) {
#pragma unroll
        for (int uu = 0; uu < NQT; ++uu) {
            const int cur = uu & 1;
            if (uu + 1 < NQT) gen_b(uu + 1, bh[cur ^ 1], bl[cur ^ 1]);
#pragma unroll
            for (int c = 0; c < NTC; ++c)
                sacc[c][uu] = __builtin_amdgcn_mfma_f32_16x16x32_f16(frag(ah[c]), bh[cur], sacc[c][uu], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < NTC; ++c)
                sacc[c][uu] = __builtin_amdgcn_mfma_f32_16x16x32_f16(frag(ah[c]), bl[cur], sacc[c][uu], 0, 0, 0);
#pragma unroll
            for (int c = 0; c < NTC; ++c)
                sacc[c][uu] = __builtin_amdgcn_mfma_f32_16x16x32_f16(frag(al[c]), bh[cur], sacc[c][uu], 0, 0, 0);
        }
        } else {
#pragma unroll
            for (int c = 0; c < NTC; ++c) sacc[c][0][0] += __builtin_bit_cast(float, ah[c][0] ^ al[c][1] ^ ah[c][2] ^ al[c][3]);
        }
    }

    // ---- flush: rows = components kbase + 64 (c / 4) + 4 (4 g + r) + c % 4 ----
    if (dbg & 4) { if (sacc[0][0][0] == 1.2345f) Sp[0] = 1.0; return; }
    const float* isx = sc + 64;
#pragma unroll
    for (int uu = 0; uu < NQT; ++uu) {
        const int q = 16 * uu + i;
        if (q >= nq) continue;
        int a, b;
        factors(uu, a, b);
        const double unscale = (a < D ? (double)isx[a] : 1.0) * (b < D ? (double)isx[b] : 1.0) /
                               (double)(1 << kRespBits);
#pragma unroll
        for (int c = 0; c < NTC; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int slot = kbase + 64 * (c >> 2) + 4 * (4 * g + r) + (c & 3);
                // slot -> component (padded slots of a group and slots past the end: none)
                const int gi = slot % G;
                if (slot < K && gi < Greal)
                    atomicAdd(Sp + (size_t)((slot / G) * Greal + gi) * nq + q,
                              (double)sacc[c][uu][r] * unscale);
            }
    }
}

// component tiles per wave: 4 (64 components).  With at most 96 statistic columns
// (D <= 40) a wave's tile leaves room for two waves per SIMD: one wave's epilogue and
// fragment arithmetic run under the other's MFMAs (128 components per wave at one
// wave per SIMD measured slower: 7.5 against 6.0 ms per 1 M frames before the loads
// were batched, and its 512 registers spill in hipcc's hands).
inline int accf_ntc(int cov, int D) { return 4; }
inline int accf_nqt(int cov, int D) { return nslab_of(cov, D) * 4 <= 96 ? 6 : 9; }

// groups of a mixture set padded to a power of two (>= 4: a lane's 4 components then
// share their state)
inline int group_pad(int G) {
    int p = 1;
    while (p < G) p <<= 1;
    return p;
}
// the fused accumulation needs a multiple of 4 only (no group reductions)
inline int accf_group_pad(int S, int G) { return (G + 3) / 4 * 4; }
inline bool supported_llh_padded(int D, int S, int G) {
    return supported_llh(D, S, S > 1 ? group_pad(G) : G);
}

inline int nt16_for(int S, int K) { return S > 1 ? 16 : (K <= 64 ? 4 : (K <= 128 ? 8 : 16)); }
inline int nchunks16_for(int S, int K) { return S > 1 ? (K + 255) / 256 : 1; }
size_t up256(size_t n) { return (n + 255) / 256 * 256; }

}  // namespace

bool supported_llh_split(int D, int S, int G) { return supported_llh_padded(D, S, G); }
// Mixture sets whose responsibilities can leave the E-step as packed tiles: full
// covariance (the two-wave kernel), groups of 4 .. 128 components, a power of two
bool supported_llh_packed_sets(int cov, int D, int S, int G) {
    return cov == BEER_FULL && S > 1 && G >= 4 && G <= 128 && (G & (G - 1)) == 0 &&
           supported_llh_padded(D, S, G) && supported_acc(D, S * G);
}

int f16_range_hazard(int64_t nframes, int D, const float* X, void* scratch, int* hazard,
                     hipStream_t s) {
    // scratch: 64 unsigned + 64 double
    unsigned* absmax = reinterpret_cast<unsigned*>(scratch);
    double* abssum = reinterpret_cast<double*>(reinterpret_cast<char*>(scratch) + 256);
    hipError_t e = hipMemsetAsync(scratch, 0, 256 + 512, s);
    if (e != hipSuccess) return -(int)e;
    int64_t blocks = (nframes + 8 * 16 - 1) / (8 * 16);
    if (blocks > 512) blocks = 512;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(512), 0, s, X, nframes, D, absmax);
    hipLaunchKernelGGL(abssum_kernel, dim3((unsigned)blocks), dim3(512), 0, s, X, nframes, D, abssum);
    hipLaunchKernelGGL(hazard_kernel, dim3(1), dim3(1), 0, s, absmax, abssum, nframes, D, hazard);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

// 129 .. 256 components: the E-step kernel (KS = 2) also leaves the transposed,
// range-scaled frames behind its tiles -- [header][R tiles][X^T tiles]
inline bool packed_has_xt(int K) { return K > kPackedComps && K <= 2 * kPackedComps; }
inline size_t packed_tiles_bytes(int64_t nframes, int K) {
    const int64_t tiles = (nframes + kPackedFrames - 1) / kPackedFrames;
    const int nblk = (K + kPackedComps - 1) / kPackedComps;
    return (size_t)tiles * nblk * kPackedComps * kPackedFrames * 4;
}
size_t packed_resps_bytes(int64_t nframes, int D, int K) {
    const int64_t tiles = (nframes + kPackedFrames - 1) / kPackedFrames;
    return kPackedHeader + packed_tiles_bytes(nframes, K) +
           (packed_has_xt(K) ? (size_t)tiles * xt_pieces(D) * kPiece : 0);
}

int pack_resps(int64_t nframes, int D, int S, int G, const float* X, const float* R,
               const float* SR, void* packed, hipStream_t s) {
    const int K = S * G;
    if ((K & 3) || D < 1 || D > 64) return BEER_EINVAL;
    if (nframes == 0) return BEER_OK;
    // header: the frame scales (the absmax scratch borrows the first tile, which the
    // pack kernel overwrites afterwards)
    float* sc = reinterpret_cast<float*>(packed);
    unsigned* tiles = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(packed) + kPackedHeader);
    const int rc = launch_scales(X, nframes, D, tiles, sc, s);
    if (rc != BEER_OK) return rc;
    const int64_t ntile = (nframes + kPackedFrames - 1) / kPackedFrames;
    const int nblk = (K + kPackedComps - 1) / kPackedComps;
    hipLaunchKernelGGL(pack_resps_kernel, dim3((unsigned)ntile, (unsigned)nblk), dim3(256), 0, s,
                       nframes, K, S, G, R, SR, tiles);
    if (packed_has_xt(K))               // what the E-step kernel would have left behind
        hipLaunchKernelGGL(xt_image_kernel, dim3((unsigned)ntile), dim3(256), 0, s, nframes, D,
                           xt_pieces(D), X, sc,
                           reinterpret_cast<float*>(reinterpret_cast<char*>(tiles) +
                                                    packed_tiles_bytes(nframes, K)));
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

int unpack_resps(int64_t nframes, int K, const void* packed, float* resps, hipStream_t s) {
    const int64_t n = nframes * K;
    if (n == 0) return BEER_OK;
    hipLaunchKernelGGL(unpack_resps_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                       nframes, K, reinterpret_cast<const unsigned*>(packed), resps);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

int frame_scales(int64_t nframes, int D, const float* X, float* scales, void* scratch,
                 hipStream_t s) {
    if (D < 1 || D > 64 || nframes < 0) return BEER_EINVAL;
    return launch_scales(X, nframes, D, reinterpret_cast<unsigned*>(scratch), scales, s);
}

size_t estep16_workspace_bytes(int cov, int D, int S, int G) {
    if (!supported_llh_padded(D, S, G)) return 0;
    if (S > 1) G = group_pad(G);
    const int K = S * G, NT = nt16_for(S, K), nchunks = nchunks16_for(S, K);
    const size_t kpad = (size_t)nchunks * NT * 16;
    return up256(((size_t)nchunks * nk16_of(cov, D) * NT + kPadBlocks) * 2048) +
           up256(kpad * sizeof(float)) + up256((size_t)(nk16_of(cov, D) + 1) * 8 * sizeof(int)) +
           1024;
}

int estep_f16x3(int cov, int64_t nframes, int D, int S, int G, const float* X, const float* expT,
                const float* logw, float* resps, float* log_norm, double* llh_sum, void* ws,
                size_t ws_bytes, hipStream_t s, bool packed, const float* given_scales,
                const float* moments) {
    if (packed && S != 1 && !supported_llh_packed_sets(cov, D, S, G)) return BEER_EINVAL;
    if (!supported_llh_padded(D, S, G) || ws_bytes < estep16_workspace_bytes(cov, D, S, G))
        return BEER_EINVAL;
    // mixture sets whose G is not a power of two: groups padded to Gp slots (logit
    // -1e30), log-normalisers only (the responsibilities would come out in the padded
    // layout)
    const int Greal = G, Kreal = S * G;
    if (S > 1) G = group_pad(G);
    if (G != Greal && resps) return BEER_EINVAL;
    const int K = S * G;
    const int NT = nt16_for(S, K), nchunks = nchunks16_for(S, K), nk = nk16_of(cov, D);
    const int kpad = nchunks * NT * 16;
    g_cov_of_launch = cov;
    char* w = reinterpret_cast<char*>(ws);
    _Float16* P = reinterpret_cast<_Float16*>(w);
    w += up256(((size_t)nchunks * nk * NT + kPadBlocks) * 2048);
    float* inv_scale = reinterpret_cast<float*>(w);
    w += up256((size_t)kpad * sizeof(float));
    int* tab = reinterpret_cast<int*>(w);
    w += up256((size_t)(nk + 1) * 8 * sizeof(int));
    unsigned* absmax = reinterpret_cast<unsigned*>(w);
    const float* sc = given_scales;               // the caller's (frame_scales), or made here
    if (!sc) {
        float* mine = reinterpret_cast<float*>(w + 256);
        const int rc = launch_scales(X, nframes, D, absmax, mine, s);
        if (rc != BEER_OK) return rc;
        sc = mine;
    }
    hipLaunchKernelGGL(pack16_kernel, dim3(kpad),
                       dim3(moments ? 256 : pack16_threads(cov, D, kpad)), pack16_lds(cov, D), s,
                       cov, D, Kreal, NT, expT, logw, sc, P, inv_scale, tab, Greal, G, moments);
    BEER_LAUNCH_CHECK();
#define BEER_LLH16(NT_, MT_, GQ_)                                                                \
    return launch_llh16<NT_, MT_, GQ_>(nframes, D, K, S, G, gl, jw, nchunks, nk, X, P, inv_scale, \
                                       sc, tab, resps, log_norm, llh_sum, s)
    if (S == 1 && packed) {
        const int gl = 16, jw = 4;
        if (hipMemcpyAsync(resps, sc, kPackedHeader, hipMemcpyDeviceToDevice, s) != hipSuccess)
            return BEER_EINVAL;
        resps += kPackedHeader / sizeof(float);
#define BEER_LLH16P(NT_, GQ_)                                                                    \
    return launch_llh16<NT_, 2, GQ_, true>(nframes, D, K, S, G, gl, jw, nchunks, nk, X, P,       \
                                           inv_scale, sc, tab, resps, log_norm, llh_sum, s)
        if (NT == 4) BEER_LLH16P(4, 1);
        if (NT == 8) BEER_LLH16P(8, 2);
        // 129 .. 256 components: two waves per frame group, 128 components each
        float* xt = reinterpret_cast<float*>(reinterpret_cast<char*>(resps) +
                                             packed_tiles_bytes(nframes, K));
        const int xtf = xt_pieces(D) * (kPiece / 4);
        if (cov == BEER_FULL)
            return launch_llh16<8, 4, 2, true, 2, false>(nframes, D, K, S, G, gl, jw, nchunks, nk, X,
                                                         P, inv_scale, sc, tab, resps, log_norm,
                                                         llh_sum, s, xt, xtf);
        return launch_llh16<8, 4, 2, true, 2>(nframes, D, K, S, G, gl, jw, nchunks, nk, X, P,
                                              inv_scale, sc, tab, resps, log_norm, llh_sum, s, xt,
                                              xtf);
#undef BEER_LLH16P
    }
    if (S == 1) {
        const int gl = 16, jw = 4;
        if (NT == 4) BEER_LLH16(4, 2, 1);
        if (NT == 8) BEER_LLH16(8, 2, 2);
        BEER_LLH16(16, 2, 4);
    }
    const int jw = G < 4 ? G : 4;
    const int gl = G < 4 ? 1 : (G < 64 ? G / 4 : 16);
    const int gq = G <= 64 ? 1 : G / 64;
    // Full covariance (a long parameter stream per frame tile): groups of at most
    // 128 components fit one wave's half of a 256-component chunk, so the waves can
    // split the components as in the packed kernel (half the stream per MFMA).
    if (cov == BEER_FULL && G <= 128 && packed) {
        // ... and hand the responsibilities over as the accumulation's LDS tiles
        if (hipMemcpyAsync(resps, sc, kPackedHeader, hipMemcpyDeviceToDevice, s) != hipSuccess)
            return BEER_EINVAL;
        resps += kPackedHeader / sizeof(float);
        if (gq == 1)
            return launch_llh16<8, 4, 1, true, 2, false, false, true>(
                nframes, D, K, S, G, gl, jw, nchunks, nk, X, P, inv_scale, sc, tab, resps, log_norm,
                llh_sum, s);
        return launch_llh16<8, 4, 2, true, 2, false, false, true>(
            nframes, D, K, S, G, gl, jw, nchunks, nk, X, P, inv_scale, sc, tab, resps, log_norm,
            llh_sum, s);
    }
    if (cov == BEER_FULL && G <= 128) {
        if (gq == 1)
            return launch_llh16<8, 4, 1, false, 2, false>(nframes, D, K, S, G, gl, jw, nchunks, nk,
                                                          X, P, inv_scale, sc, tab, resps,
                                                          log_norm, llh_sum, s);
        return launch_llh16<8, 4, 2, false, 2, false>(nframes, D, K, S, G, gl, jw, nchunks, nk, X,
                                                      P, inv_scale, sc, tab, resps, log_norm,
                                                      llh_sum, s);
    }
    if (!resps && jw == 4) {
        // log-normalisers only (the accumulation recomputes the responsibilities)
#define BEER_LLH16N(GQ_)                                                                          \
    return launch_llh16<16, 2, GQ_, false, 1, true, true>(nframes, D, K, S, G, gl, jw, nchunks,   \
                                                          nk, X, P, inv_scale, sc, tab, resps,    \
                                                          log_norm, llh_sum, s)
        switch (gq) {
            case 1: BEER_LLH16N(1);
            case 2: BEER_LLH16N(2);
            default: BEER_LLH16N(4);
        }
#undef BEER_LLH16N
    }
    switch (gq) {
        case 1: BEER_LLH16(16, 2, 1);
        case 2: BEER_LLH16(16, 2, 2);
        default: BEER_LLH16(16, 2, 4);
    }
#undef BEER_LLH16
}

size_t acc16_workspace_bytes(int cov, int D, int K) {
    if (!supported_acc(D, K)) return 0;
    const int nslab = nslab_of(cov, D);
    return up256((size_t)K * nslab * 4 * sizeof(double)) + up256((size_t)nslab * sizeof(int)) + 1024;
}

size_t acc16p_workspace_bytes(int cov, int64_t nframes, int D, int K) {
    const size_t base = acc16_workspace_bytes(cov, D, K);
    if (base == 0) return 0;
    const int64_t tiles = (nframes + kA16FT - 1) / kA16FT;
    return base + (size_t)tiles * xt_pieces(D) * kPiece;
}

// ... with state posteriors multiplied in by the accumulation kernel: S states of G
// components (a power of two, 8 .. 128), more than 16 statistic tiles, X^T tiles of at
// most 3 pieces (D <= 43: three LDS buffers have to fit)
bool supported_acc_sets(int cov, int D, int S, int G) {
    return S >= 1 && G >= 8 && G <= 128 && (G & (G - 1)) == 0 && supported_acc(D, S * G) &&
           (nslab_of(cov, D) * 4 + 15) / 16 > 16 && xt_pieces(D) <= 3;
}
inline int acc_sets_spad(int S, int G) {
    return (S * G + kPackedComps - 1) / kPackedComps * (kPackedComps / G);
}
size_t acc16s_workspace_bytes(int cov, int64_t nframes, int D, int S, int G) {
    if (!supported_acc_sets(cov, D, S, G)) return 0;
    const int64_t tiles = (nframes + kA16FT - 1) / kA16FT;
    return up256(acc16p_workspace_bytes(cov, nframes, D, S * G)) +
           (size_t)tiles * acc_sets_spad(S, G) * kA16FT * sizeof(float) + 1024;
}

int acc_f16x3_packed(int cov, int64_t nframes, int D, int K, const float* X, const void* Rimg,
                     double* acc, void* ws, size_t ws_bytes, hipStream_t s, int S, int G,
                     const float* SR) {
    if (!supported_acc(D, K) || ws_bytes < acc16p_workspace_bytes(cov, nframes, D, K))
        return BEER_EINVAL;
    if (SR && (S * G != K || !supported_acc_sets(cov, D, S, G) ||
               ws_bytes < acc16s_workspace_bytes(cov, nframes, D, S, G)))
        return BEER_EINVAL;
    const int nslab = nslab_of(cov, D), nq = nslab * 4;
    char* w = reinterpret_cast<char*>(ws);
    double* Sp = reinterpret_cast<double*>(w);
    w += up256((size_t)K * nq * sizeof(double));
    int* tab = reinterpret_cast<int*>(w);
    // frame scales: the header K1 left in front of the tiles
    const float* sc = reinterpret_cast<const float*>(Rimg);
    Rimg = reinterpret_cast<const char*>(Rimg) + kPackedHeader;
    float* Xt = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + acc16_workspace_bytes(cov, D, K));
    hipError_t e = hipMemsetAsync(Sp, 0, (size_t)K * nq * sizeof(double), s);
    if (e != hipSuccess) return -(int)e;
    hipLaunchKernelGGL(tab_kernel, dim3(1), dim3(256), 0, s, cov, D, tab);
    BEER_LAUNCH_CHECK();
    const int64_t tiles = (nframes + kA16FT - 1) / kA16FT;
    const int NX = xt_pieces(D);
    if (packed_has_xt(K) && !SR) {           // the E-step kernel left the image behind the tiles
                                             // (one mixture; the kernel of a set does not)
        Xt = reinterpret_cast<float*>(const_cast<char*>(reinterpret_cast<const char*>(Rimg)) +
                                      packed_tiles_bytes(nframes, K));
    } else {
        hipLaunchKernelGGL(xt_image_kernel, dim3((unsigned)tiles), dim3(256), 0, s, nframes, D, NX,
                           X, sc, Xt);
        BEER_LAUNCH_CHECK();
    }
    const int ntiles = (nq + 15) / 16;
    static const int form = [] { const char* e = getenv("BEER_ACC16P"); return e ? atoi(e) : 2; }();
    float* Gt = nullptr;
    int lgG = 0;
    if (SR) {
        Gt = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) +
                                      up256(acc16p_workspace_bytes(cov, nframes, D, K)));
        const int spad = acc_sets_spad(S, G);
        hipLaunchKernelGGL(gt_image_kernel, dim3((unsigned)tiles, (unsigned)((spad + 63) / 64)),
                           dim3(256), 0, s, nframes, S, spad, SR, Gt);
        BEER_LAUNCH_CHECK();
        while ((1 << lgG) < G) ++lgG;
    }
    if (SR || (form == 2 && ntiles > 16)) {
        // second form: 4 statistic tiles per wave, LDS-DMA staging (see acc16d_kernel)
        const int gx = (ntiles + 31) / 32;
        const int gy = (K + 16 * kA16MC - 1) / (16 * kA16MC);
        int64_t gz = (256 + (int64_t)gx * gy - 1) / ((int64_t)gx * gy);
        const int64_t max_z = (nframes + 1023) / 1024,
                      min_z = (nframes + kA16MaxFrames - 1) / kA16MaxFrames;
        if (gz > max_z) gz = max_z;
        if (gz < min_z) gz = min_z;
        if (gz < 1) gz = 1;
        int64_t fpb = (nframes + gz - 1) / gz;
        fpb = (fpb + kA16FT - 1) / kA16FT * kA16FT;
        gz = (nframes + fpb - 1) / fpb;
        const size_t lds = SR ? 3 * ((size_t)(NX + 8) * kPiece + 4096)
                              : 2 * (size_t)(NX + 8) * kPiece;
        const int64_t nyz = ((int64_t)gy * gz + 7) / 8 * 8;
        const dim3 grid((unsigned)(nyz * gx));
#define BEER_ACC16D(NX_, SR_)                                                                    \
    do {                                                                                         \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(acc16d_kernel<NX_, SR_>),        \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);         \
        hipLaunchKernelGGL((acc16d_kernel<NX_, SR_>), grid, dim3(512), lds, s, nframes, D, K,    \
                           nslab, Xt, reinterpret_cast<const unsigned*>(Rimg), tab, sc, fpb, Sp, \
                           gx, gy, (int)gz, Gt, lgG);                                            \
    } while (0)
        if (SR) {
            if (NX == 1) BEER_ACC16D(1, true);
            else if (NX == 2) BEER_ACC16D(2, true);
            else BEER_ACC16D(3, true);
        }
        else if (NX == 1) BEER_ACC16D(1, false);
        else if (NX == 2) BEER_ACC16D(2, false);
        else if (NX == 3) BEER_ACC16D(3, false);
        else if (NX == 4) BEER_ACC16D(4, false);
        else BEER_ACC16D(5, false);
#undef BEER_ACC16D
        BEER_LAUNCH_CHECK();
        const int64_t total2 = (int64_t)K * stats_dim(cov, D);
        hipLaunchKernelGGL(unpack_kernel, dim3((unsigned)((total2 + 255) / 256)), dim3(256), 0, s,
                           cov, D, K, Sp, acc);
        BEER_LAUNCH_CHECK();
        return BEER_OK;
    }
    // 8 waves, two per SIMD (256 registers each): measured faster than 4 waves with
    // twice the statistic tiles each (1.26 against 1.31 ms at K = 256, D = 40, full)
    const int waves = 8;
    const int NQ = ntiles > waves ? 2 : 1;
    const int gx = (ntiles + NQ * waves - 1) / (NQ * waves);
    const int gy = (K + 16 * kA16MC - 1) / (16 * kA16MC);
    // two workgroups per CU: fewer, longer frame chunks cost less prologue and fewer
    // fp64 atomics than the 4 per CU of acc_f16x3 (1.25 -> 1.21 ms at config 2)
    int64_t gz = (512 + (int64_t)gx * gy - 1) / ((int64_t)gx * gy);
    const int64_t max_z = (nframes + 1023) / 1024, min_z = (nframes + kA16MaxFrames - 1) / kA16MaxFrames;
    if (gz > max_z) gz = max_z;
    if (gz < min_z) gz = min_z;
    if (gz < 1) gz = 1;
    int64_t fpb = (nframes + gz - 1) / gz;
    fpb = (fpb + kA16FT - 1) / kA16FT * kA16FT;
    gz = (nframes + fpb - 1) / fpb;
    const size_t lds = 2 * (size_t)(NX + 8) * kPiece;
    const int64_t nyz = ((int64_t)gy * gz + 7) / 8 * 8;
    const dim3 grid((unsigned)(nyz * gx));
#define BEER_ACC16P(NQ_, NX_, W_)                                                                \
    do {                                                                                         \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(acc16p_kernel<NQ_, NX_, W_>),    \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);         \
        hipLaunchKernelGGL((acc16p_kernel<NQ_, NX_, W_>), grid, dim3(64 * W_), lds, s, nframes,  \
                           D, K, nslab, Xt, reinterpret_cast<const unsigned*>(Rimg), tab, sc,    \
                           fpb, Sp, gx, gy, (int)gz);                                            \
    } while (0)
#define BEER_ACC16P_NX(NQ_, W_)                                                                  \
    do {                                                                                         \
        if (NX == 1) BEER_ACC16P(NQ_, 1, W_);                                                    \
        else if (NX == 2) BEER_ACC16P(NQ_, 2, W_);                                               \
        else if (NX == 3) BEER_ACC16P(NQ_, 3, W_);                                               \
        else if (NX == 4) BEER_ACC16P(NQ_, 4, W_);                                               \
        else BEER_ACC16P(NQ_, 5, W_);                                                            \
    } while (0)
    if (NQ == 2) BEER_ACC16P_NX(2, 8);
    else BEER_ACC16P_NX(1, 8);
#undef BEER_ACC16P_NX
#undef BEER_ACC16P
    BEER_LAUNCH_CHECK();
    const int64_t total = (int64_t)K * stats_dim(cov, D);
    hipLaunchKernelGGL(unpack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, cov,
                       D, K, Sp, acc);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

// statistics per Gaussian small enough for a wave's register tile: diagonal and
// isotropic covariances up to D = 64 (nq <= 144)
bool supported_accf(int cov, int D, int S, int G) {
    return cov != BEER_FULL && D >= 1 && D <= 64 && S >= 1 && G >= 1 && G <= 256 &&
           nslab_of(cov, D) * 4 <= 144;
}

size_t accf_workspace_bytes(int cov, int D, int S, int G) {
    if (!supported_accf(cov, D, S, G)) return 0;
    const int Kreal = S * G;
    G = accf_group_pad(S, G);
    const int K = S * G, NTC = accf_ntc(cov, D), nk = nk16_of(cov, D);
    const int nchunks = (K + 16 * NTC - 1) / (16 * NTC), nq = nslab_of(cov, D) * 4;
    return up256(((size_t)nchunks * nk * NTC + kPadBlocks) * 2048) +
           up256((size_t)nchunks * NTC * 16 * sizeof(float)) +
           up256((size_t)(nk + 1) * 8 * sizeof(int)) + 1024 +
           up256((size_t)Kreal * nq * sizeof(double));
}

int acc_fused_f16x3(int cov, int64_t nframes, int D, int S, int G, const float* X,
                    const float* expT, const float* logw, const float* log_norm,
                    const float* sr, double* acc, void* ws, size_t ws_bytes, hipStream_t s) {
    if (!supported_accf(cov, D, S, G) || ws_bytes < accf_workspace_bytes(cov, D, S, G))
        return BEER_EINVAL;
    // component slots: groups padded to a multiple of 4 (one state per lane's 4
    // components); the statistics image Sp stays in the components' own order
    const int Greal = G, Kreal = S * G;
    G = accf_group_pad(S, G);
    const int K = S * G, NTC = accf_ntc(cov, D), NQT = accf_nqt(cov, D), nk = nk16_of(cov, D);
    const int nchunks = (K + 16 * NTC - 1) / (16 * NTC), kpad = nchunks * NTC * 16;
    const int nslab = nslab_of(cov, D), nq = nslab * 4;
    char* w = reinterpret_cast<char*>(ws);
    _Float16* P = reinterpret_cast<_Float16*>(w);
    w += up256(((size_t)nchunks * nk * NTC + kPadBlocks) * 2048);
    float* inv_scale = reinterpret_cast<float*>(w);
    w += up256((size_t)kpad * sizeof(float));
    int* tab = reinterpret_cast<int*>(w);
    w += up256((size_t)(nk + 1) * 8 * sizeof(int));
    unsigned* absmax = reinterpret_cast<unsigned*>(w);
    float* sc = reinterpret_cast<float*>(w + 256);
    w += 1024;
    double* Sp = reinterpret_cast<double*>(w);
    hipError_t e = hipMemsetAsync(Sp, 0, (size_t)Kreal * nq * sizeof(double), s);
    if (e != hipSuccess) return -(int)e;
    const int rc = launch_scales(X, nframes, D, absmax, sc, s);
    if (rc != BEER_OK) return rc;
    hipLaunchKernelGGL(pack16_kernel, dim3(kpad), dim3(pack16_threads(cov, D, kpad)), pack16_lds(cov, D), s, cov, D, Kreal, NTC, expT, logw,
                       sc, P, inv_scale, tab, Greal, G);
    BEER_LAUNCH_CHECK();
    // waves per workgroup: 8 (two per SIMD) with 64-component chunks, 4 with 128
    const int waves = (NTC == 4 && NQT == 6) ? 8 : 4;
    // frames per workgroup: <= kAfMaxFramesPerWave per wave, about one workgroup of
    // 8 waves (two of 4) per CU and round
    static const int rounds = [] { const char* e = getenv("BEER_ACCF_ROUNDS"); return e ? atoi(e) : 6; }();
    int64_t gz = ((waves == 8 ? 256 : 512) * rounds + nchunks - 1) / nchunks;
    gz = (gz + 7) / 8 * 8;                       // whole rows of the XCD-aware grid
    const int64_t min_z = (nframes + (int64_t)waves * kAfMaxFramesPerWave - 1) /
                          ((int64_t)waves * kAfMaxFramesPerWave);
    const int64_t max_z = (nframes + 32 * waves - 1) / (32 * waves);
    if (gz > max_z) gz = max_z;
    if (gz < min_z) gz = min_z;
    if (gz < 1) gz = 1;
    int64_t fpb = (nframes + gz - 1) / gz;
    fpb = (fpb + 32 * waves - 1) / (32 * waves) * (32 * waves);
    gz = (nframes + fpb - 1) / fpb;
    const size_t lds = (size_t)nk * NTC * 2048 + (size_t)(nk + 1) * 8 * sizeof(int) + 256 +
                       (size_t)waves * (32 * ld16_of(D) + (D + 2) * kAfXS) * sizeof(float);
    const dim3 grid(xcd_grid(gz, nchunks, nchunks));
    const bool g4 = (G % 4) == 0;
    static const int dbg = [] { const char* e = getenv("BEER_ACCF_DBG"); return e ? atoi(e) : 0; }();
#define BEER_ACCF(NTC_, NQT_, G4_, W_)                                                           \
    do {                                                                                         \
        constexpr int XP_ = NQT_ == 6 ? 5 : 8;      /* D <= 40 <=> C4 <= 10 <=> nq <= 96 */       \
        (void)hipFuncSetAttribute(                                                               \
            reinterpret_cast<const void*>(accf_kernel<NTC_, NQT_, G4_, W_, XP_>),                \
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                               \
        hipLaunchKernelGGL((accf_kernel<NTC_, NQT_, G4_, W_, XP_>), grid, dim3(64 * W_), lds, s, \
                           nframes, D, K, S, G, Greal, nk, nslab, X, P, inv_scale, sc, tab,      \
                           log_norm, sr, fpb, Sp, dbg);                                          \
    } while (0)
    if (NQT == 6) { if (g4) BEER_ACCF(4, 6, true, 8); else BEER_ACCF(4, 6, false, 8); }
    else { if (g4) BEER_ACCF(4, 9, true, 4); else BEER_ACCF(4, 9, false, 4); }
#undef BEER_ACCF
    BEER_LAUNCH_CHECK();
    const int64_t total = (int64_t)Kreal * stats_dim(cov, D);
    hipLaunchKernelGGL(unpack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, cov,
                       D, Kreal, Sp, acc);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

int acc_f16x3(int cov, int64_t nframes, int D, int S, int G, const float* X, const float* R,
              const float* SR, double* acc, void* ws, size_t ws_bytes, hipStream_t s) {
    const int K = S * G;
    if (!supported_acc(D, K) || ws_bytes < acc16_workspace_bytes(cov, D, K)) return BEER_EINVAL;
    const int nslab = nslab_of(cov, D), nq = nslab * 4;
    char* w = reinterpret_cast<char*>(ws);
    double* Sp = reinterpret_cast<double*>(w);
    w += up256((size_t)K * nq * sizeof(double));
    int* tab = reinterpret_cast<int*>(w);
    w += up256((size_t)nslab * sizeof(int));
    unsigned* absmax = reinterpret_cast<unsigned*>(w);
    float* sc = reinterpret_cast<float*>(w + 256);
    hipError_t e = hipMemsetAsync(Sp, 0, (size_t)K * nq * sizeof(double), s);
    if (e != hipSuccess) return -(int)e;
    const int rc = launch_scales(X, nframes, D, absmax, sc, s);
    if (rc != BEER_OK) return rc;
    hipLaunchKernelGGL(tab_kernel, dim3(1), dim3(256), 0, s, cov, D, tab);
    BEER_LAUNCH_CHECK();
    const int ntiles = (nq + 15) / 16;
    const int waves = kA16Threads / 64;
    const int NQ = ntiles > 2 * waves ? 4 : (ntiles > waves ? 2 : 1);
    const int gx = (ntiles + NQ * waves - 1) / (NQ * waves);
    const int mc = SR && NQ > 2 ? kA16MCsr : kA16MC;
    const int gy = (K + 16 * mc - 1) / (16 * mc);
    int64_t gz = (1024 + (int64_t)gx * gy - 1) / ((int64_t)gx * gy);
    const int64_t max_z = (nframes + 1023) / 1024, min_z = (nframes + kA16MaxFrames - 1) / kA16MaxFrames;
    if (gz > max_z) gz = max_z;
    if (gz < min_z) gz = min_z;
    if (gz < 1) gz = 1;
    int64_t fpb = (nframes + gz - 1) / gz;
    fpb = (fpb + kA16FT - 1) / kA16FT * kA16FT;
    gz = (nframes + fpb - 1) / fpb;
    const size_t lds = 2 * ((size_t)(D + 3) * kA16XS * 4 + (size_t)16 * mc * kA16RS * 2 * 2) + 256;
    const int64_t nyz = ((int64_t)gy * gz + 7) / 8 * 8;
    const dim3 grid((unsigned)(nyz * gx));
#define BEER_ACC16(NQ_, SR_)                                                                     \
    do {                                                                                         \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(acc16_kernel<NQ_, SR_>),         \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);         \
        hipLaunchKernelGGL((acc16_kernel<NQ_, SR_>), grid, dim3(kA16Threads), lds, s, nframes,   \
                           D, K, G, S, nslab, X, R, SR, tab, sc, fpb, Sp, gx, gy, (int)gz);      \
    } while (0)
    if (SR) {
        if (NQ == 4) BEER_ACC16(4, true);
        else if (NQ == 2) BEER_ACC16(2, true);
        else BEER_ACC16(1, true);
    } else {
        if (NQ == 4) BEER_ACC16(4, false);
        else if (NQ == 2) BEER_ACC16(2, false);
        else BEER_ACC16(1, false);
    }
#undef BEER_ACC16
    BEER_LAUNCH_CHECK();
    const int64_t total = (int64_t)K * stats_dim(cov, D);
    hipLaunchKernelGGL(unpack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, cov,
                       D, K, Sp, acc);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

}  // namespace beer_mfma
