// fp32 E-step on the bf16 matrix pipes ("bf16x3").
//
// gfx950 runs v_mfma_f32_16x16x4_f32 at 1/16 of the rate of
// v_mfma_f32_16x16x32_bf16 (MI355X_MICROARCH.md: 157 TF vs 2.5 PF dense).  Both
// GEMMs of the E-step are therefore evaluated with every fp32 operand held
// EXACTLY as three bf16 pieces, v = p0 + p1 + p2 (3 x 8 significand bits = the 24
// of fp32; bf16 has fp32's exponent range: no scaling, no range hazards), and the
// six leading partial products of each multiplication, accumulated in fp32:
//
//     a * b  ~=  a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0)
//
// What is dropped (a1 b2 + a2 b1 + a2 b2) is below 2^-23 |a b| in the worst case
// and 2^-25 |a b| typically, with no systematic sign (the pieces are rounded to
// nearest): the operands are fp32's own, each product is as accurate as an fp32
// multiply, the accumulation IS fp32 -- 6 MFMAs of 16 cycles per 32-deep step
// against 8 MFMAs of 32 cycles on the fp32 pipe: 2.7x its rate.
//
// Same slab enumeration, component interleave and softmax epilogue as
// estep_mfma.hip; one k-step of the bf16 MFMA (32 deep) covers 8 slabs, lane
// k-block g (8 values) = slabs 8s+2g and 8s+2g+1.
//
// Operand mapping of v_mfma_f32_16x16x32_bf16 (lane l: i = l & 15, g = l >> 4):
// A[i][k = 8g..8g+7], B[k = 8g..8g+7][n = i], C/D row 4g + r, column i.
//
// Reference restated: beer/dists/normalwishart.py:30-38, 88-92,
// beer/models/mixture.py:79-101, beer/models/mixtureset.py:85-112,
// beer/models/normalset.py:117-123.

#include <cstdlib>
#include <type_traits>
#include <utility>

#include "estep_mfma.h"
#include "estep_tiles.h"

// Timing experiments (tools/ab_build.sh; all but 0 give wrong results):
#ifndef BEER_K1_ABL
#define BEER_K1_ABL 0      // K1: 1 = no fragment arithmetic, 2 = and no parameter loads, 3 = and no epilogue
#endif
#ifndef BEER_K2_ABL
#define BEER_K2_ABL 0      // K2: 1 = no B fragments, 2 = and no A loads, 3 = and no atomics
#endif
#ifndef BEER_AF_ABL
#define BEER_AF_ABL 0      // fused accumulation, bits: 1 no flush, 8 no exp / split, 16 no statistics B fragments,
#endif                     // 32 no logit A fragments, 64 no tile skipping
#ifndef BEER_AFI_ABL
#define BEER_AFI_ABL 0     // accfi_kernel, bits: 1 one B fragment load per tile, 2 no flush, 4 one A fragment load per tile
#endif
#ifndef BEER_LNFI_ABL
#define BEER_LNFI_ABL 0     // lnfi_kernel, bits: 1 no epilogue, 2 no MFMAs, 4 no LDS reads of B, 8 no A loads
#endif
#ifndef BEER_LNFI_SLEEP
#define BEER_LNFI_SLEEP 0   // lnfi_kernel: s_sleep count (x 64 cycles) of the second wave of every SIMD at its start
#endif
#ifndef BEER_ACCFI_SLEEP
#define BEER_ACCFI_SLEEP 0  // accfi_kernel: the same
#endif
#ifndef BEER_K1_FENCE
#define BEER_K1_FENCE 0    // K1: scheduling fence every n MFMAs of the hand-placed stream (0 = none)
#endif

namespace beer_mfma {

namespace {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// compile-time loops / indices for the hand-placed instruction streams (everything that
// selects a register must be a constant when the code is generated, not after unrolling)
template <int V> using ic = std::integral_constant<int, V>;
template <int... I, typename F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(ic<I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

constexpr int NP = kPackedPieces;      // pieces per operand
// the six products (piece of A, piece of B), by decreasing weight
constexpr int kProdA[6] = {0, 0, 1, 0, 1, 2};
constexpr int kProdB[6] = {0, 1, 0, 2, 1, 0};

__device__ __forceinline__ bf8 as_bf8(const u4& w) { return __builtin_bit_cast(bf8, w); }
__device__ __forceinline__ f32x4 mfma_bf16(const u4& a, const u4& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf8(a), as_bf8(b), c, 0, 0, 0);
}
// The same MFMA with its accumulator tied IN PLACE to an AGPR quad.  A kernel with 256
// accumulators (64 x 256 or 128 x 128 per wave, one wave per SIMD) fills the AGPR half
// of the register file exactly; given the builtin, hipcc's allocator treats
// accumulators and fragments as one "any vector register" class and shuffles them
// between the two halves -- 0.55 v_accvgpr_* per MFMA and, in K2, 110 scratch accesses
// per tile pair (tools/isa_stats.py).  With "+a" / "v" constraints nothing moves.  What
// hipcc then no longer does is pad hazards around the instruction
// (cdna_hip_programming.md section 5.7): FIRST = the fragment operands may have been
// written by the VALU instruction just before (2 wait states), and the accumulators
// must not be read before mfma_drain().  Dependent MFMAs of an accumulation chain need
// none.
template <bool FIRST>
__device__ __forceinline__ void mfma_bf16_pinned(f32x4& acc, const u4& a, const u4& b) {
    if (FIRST)
        asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    else
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// ... and with the accumulator in a VGPR quad (accumulators that VALU code reads next)
template <bool FIRST>
__device__ __forceinline__ void mfma_bf16_pinned_v(f32x4& acc, const u4& a, const u4& b) {
    if (FIRST)
        asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    else
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
// after the last pinned MFMA, before anything reads an accumulator
__device__ __forceinline__ void mfma_drain() {
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// u4 per 32-frame tile of a frame fragment image (frame_image_kernel): A fragments of
// nk_used k-steps (3 pieces x 2 frame tiles), then B fragments of nqt statistic tiles
__host__ __device__ inline size_t frame_image_tile_u4(int nk_used, int nqt) {
    return (size_t)(nk_used * NP * 2 + nqt * NP) * 64;
}
// Statistic tiles of a frame image: the DATA slabs of a diagonal / isotropic model only
// (squares and linear terms: 2 D4 <= 20 slabs = 5 tiles of 16 columns at D <= 40), in the
// table's order without its constant slabs -- data slab j is slab j + j / 7 (every eighth
// slab closes a k-step with a constant, estep_tiles.h: diag_walk).  The counts N_k, the only
// statistic a constant slab carries, are summed on the vector ALU (accfi_kernel): a sixth
// tile for one useful column in sixteen cost 24 of a wave-tile's 288 MFMAs.
// Up to 3 k-steps (D <= 40) the image carries 5 tiles; the 4-k-step format of D = 41 .. 48
// (22 / 24 data slabs: the recipes' 13 MFCCs + energy with deltas are 42 dimensions) 6.
__host__ __device__ constexpr int img_nqt(int nk_used) { return nk_used <= 3 ? 5 : 6; }
constexpr int kImgMaxNK = 4;
__host__ __device__ inline int stat_slab(int j) { return j + j / 7; }
// k-steps (8 slabs each), padded to an even count: the K1 loop is unrolled by two
__host__ __device__ inline int nk16_of(int cov, int D) {
    return ((nslab_of(cov, D) + 7) / 8 + 1) / 2 * 2;
}
// Row stride (floats) of K1's LDS frame tile: the D values, the constants 1 and
// 2^-24 and at least 6 zeros (the padding slabs read them), with stride / 4 odd --
// the 16 rows of an A-fragment ds_read_b128 then start in 16 different 16-byte
// slots of the 256-byte bank row.
__host__ __device__ inline int ld16_of(int D) {
    const int ld = 4 * d4_of(D) + 8;
    return (ld / 4) % 2 ? ld : ld + 4;
}
// A frame row carries 2^-24 next to its 1: the constant term of a component
// (-.5 E[.] + .5 E[.] - D/2 ln 2 pi + ln w, a sum of the size of the logit itself) is
// stored as fp32 value + remainder * 2^24 in the second entry of the constant slab,
// 48 bits in all -- its own rounding (up to 4e-6 at |c| ~ 100, the same for every
// frame of the component) would otherwise be the largest systematic term left.
constexpr float kConstEps = 5.9604644775390625e-8f;      // 2^-24
constexpr int kBlockU4 = NP * 64;      // u4 per (k-step, tile) block of the P image: 3 KiB

// Value of contraction entry (slab, e) for component k (the logic of pack_kernel in
// estep_mfma.hip); the constant slabs are filled by the caller (const_share).
__device__ inline double entry_value(int cov, int D, int K, int k, int slab, int e,
                                     const float* __restrict__ row, bool* is_const) {
    const int D4 = d4_of(D), Dp = 4 * D4, nslab = nslab_of(cov, D);
    *is_const = false;
    if (slab >= nslab) return 0.0;
    const int t = slab_entry(cov, D, slab);
    const int a = t & 0xff, b = ((t >> 8) & 0xff) + e, sq = t >> 16;
    if (a == Dp && b - e == Dp) {
        *is_const = true;
        return 0.0;
    }
    if (k >= K) return 0.0;
    if (sq) return b < D ? -0.5 * (double)row[cov == BEER_ISO ? D : D + b] : 0.0;
    if (a < D) {
        if (b >= D || b < a) return 0.0;
        return b == a ? -0.5 * (double)row[D + a * D + a]
                      : -0.5 * ((double)row[D + a * D + b] + (double)row[D + b * D + a]);
    }
    return b < D ? (double)row[b] : 0.0;
}

// c0[0] = the largest constant term of any component (-1/2 E[.] + 1/2 E[.] - D/2 ln 2 pi
// + ln w): at a freshly initialised model the constants (log-determinants ...) are two
// thirds of a logit and nearly the same for every component.  A softmax does not see a
// common offset, the accumulators' rounding does (it happens at the size of the
// running sum): the packed image carries every constant minus c0 and the epilogue
// adds c0 back to the log-normalisers.
__global__ __launch_bounds__(256) void const_max_kernel(int cov, int D, int K,
                                                        const float* __restrict__ E,
                                                        const float* __restrict__ logw,
                                                        float* __restrict__ c0) {
    __shared__ double red[8];
    const int Q = stats_dim(cov, D);
    double m = -1.0e300;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const double v = const_total(cov, D, E + (size_t)k * Q, logw ? (double)logw[k] : 0.0);
        if (v == v && v > m && v < 1.0e300) m = v;
    }
    m = block_max(m, red);
    if (threadIdx.x == 0) c0[0] = m > -1.0e300 ? (float)m : 0.f;
}

// One workgroup per (padded) component: its three bf16 planes at
// P[chunk][kstep][tile][piece][64 lanes x 8] and the slab table.  Component SLOT
// blockIdx.x of the image is component (slot / Gp) * G + slot % Gp when slot % Gp < G,
// else padding (G <= Gp: the groups of a mixture set padded so that any number of
// components per state runs on the group-aligned kernels; G == Gp: slots are
// components).  Padded slots carry the logit -1e30.
__global__ __launch_bounds__(256) void packx_kernel(int cov, int D, int K, int NT,
                                                    const float* __restrict__ E,
                                                    const float* __restrict__ logw,
                                                    unsigned short* __restrict__ P,
                                                    int* __restrict__ tab, int G, int Gp,
                                                    const float* __restrict__ c0, int lane_major) {
    extern __shared__ __attribute__((aligned(16))) char pack_lds[];
    const int nk = nk16_of(cov, D), nent = nk * 32;
    const int slot = blockIdx.x;
    const int k = slot % Gp < G ? (slot / Gp) * G + slot % Gp : K;     // K: a padded slot
    const int chunk = slot / (NT * 16), kk = slot % (NT * 16);
    // column i of tile c: 4 consecutive slots per lane and block of 64, or (lane_major) NT
    // consecutive slots per lane -- see lognorm_epilogue_lane_major
    const int c = lane_major ? kk % NT : 4 * (kk / 64) + (kk % 4);
    const int i = lane_major ? kk / NT : (kk % 64) / 4;
    if (slot == 0)
        for (int s = threadIdx.x; s < (nk + 1) * 8; s += blockDim.x) {
            // padding slabs read the zero columns behind the constants of a frame row
            const int Dp = 4 * d4_of(D);
            tab[s] = s < nslab_of(cov, D) ? slab_entry(cov, D, s) : ((Dp + 2) | ((Dp + 4) << 8));
        }
    // the component's expected statistics, staged once (the entries are gathered from
    // all over the row: from global memory every gather was a dependent L2 round trip)
    const int Qs = stats_dim(cov, D);
    float* rowl = reinterpret_cast<float*>(pack_lds);
    if (k < K)
        for (int q = threadIdx.x; q < Qs; q += blockDim.x) rowl[q] = E[(size_t)k * Qs + q];
    __syncthreads();
    unsigned short* base = P + ((size_t)chunk * nk * NT) * (NP * 512);
    auto put = [&](int q, float v) {
        const int s = q / 32, g = (q % 32) / 8, j = q % 8;
        unsigned short* dst = base + ((size_t)s * NT + c) * (NP * 512) + (g * 16 + i) * 8 + j;
        unsigned short p[3];
        split3_scalar(v, p);
        dst[0] = p[0];
        dst[512] = p[1];
        dst[1024] = p[2];
    };
    const int nslab = nslab_of(cov, D);
    for (int q = threadIdx.x; q < nent; q += blockDim.x) {
        bool is_const;
        const double v = entry_value(cov, D, K, k, q / 4, q % 4, rowl, &is_const);
        if (!is_const) {
            put(q, (float)v);
        } else if (q % 4 == 0) {
            // value + remainder * 2^24: entries (constant slab, 0) and (constant slab, 1);
            // a padded component gets its -1e30 in the final constant slab
            const bool last = q / 4 == nslab - 1;
            const double want = k < K ? const_share(cov, D, q / 4, rowl, logw ? (double)logw[k] : 0.0) -
                                            (last ? (double)c0[0] : 0.0)
                                      : (last ? kPadLogit : 0.0);
            const float hi = (float)want;
            const double rem = (want - (double)hi) * (1.0 / (double)kConstEps);
            put(q, hi);
            put(q + 1, (rem == rem && fabs(rem) < 1.0e30) ? (float)rem : 0.f);
            put(q + 2, 0.f);
            put(q + 3, 0.f);
        }
    }
}

// ---------------------------------------------------------------------------
// Several component chunks over the same frames: a 1-D grid whose blocks are dealt to
// the 8 XCDs round-robin, laid out so that the chunk blocks of one frame block are
// neighbours on ONE XCD (they read the same frames at about the same time: one L2
// miss, the others hit -- with a (frames, chunks) grid every chunk pass streamed the
// frames from HBM again).  The chunks are taken `cg` at a time (all frame blocks for
// chunks 0 .. cg-1, then the next cg): the packed parameters of the cg chunks an XCD
// works on must stay in its 4 MiB L2.  Grid size xcd_grid(nx, ny, cg); false =
// padding block.
// ---------------------------------------------------------------------------
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

// Rows of a frame tile from global memory into a wave's LDS image: FW rows, LPR =
// 64 / FW lanes per row, a lane takes the 16-byte pieces h, h + LPR, .. of its row.
// All loads are issued back to back on clamped piece numbers (a load behind a branch
// is issued and waited for on its own); rows past the end are zeros.
template <int FW>
__device__ __forceinline__ void stage_rows(const float* __restrict__ X, int64_t fb,
                                           int64_t nframes, int D, int LD, int lane,
                                           float* __restrict__ xw) {
    const int Dp = 4 * d4_of(D);
    if ((D & 3) == 0) {
        constexpr int LPR = 64 / FW, NPC = kMaxDimX / 4 / LPR;     // pieces per lane
        const int C4 = D >> 2, r = lane & (FW - 1), h = lane / FW;
        const int64_t f = fb + r;
        const bool valid = f < nframes;
        const f32x4* src = reinterpret_cast<const f32x4*>(X) + (valid ? f : nframes - 1) * C4;
        f32x4 xv[NPC];
#pragma unroll
        for (int it = 0; it < NPC; ++it) {
            const int pc = h + LPR * it;
            xv[it] = src[pc < C4 ? pc : C4 - 1];
        }
#pragma unroll
        for (int it = 0; it < NPC; ++it) {
            const int pc = h + LPR * it;
            if (pc < C4)
                *reinterpret_cast<f32x4*>(xw + r * LD + 4 * pc) =
                    valid ? xv[it] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // columns D .. LD - 1: the constants 1, 2^-24 at Dp (= D), zeros behind them
        for (int idx = lane; idx < FW * ((LD - D) / 4); idx += 64) {
            const int rr = idx / ((LD - D) / 4), hh = idx - rr * ((LD - D) / 4);
            *reinterpret_cast<f32x4*>(xw + rr * LD + D + 4 * hh) =
                f32x4{hh == 0 ? 1.f : 0.f, hh == 0 ? kConstEps : 0.f, 0.f, 0.f};
        }
    } else {
        for (int idx = lane; idx < FW * LD; idx += 64) {
            const int r = idx / LD, c = idx - r * LD;
            const int64_t f = fb + r;
            float v = 0.f;
            if (c < D) { if (f < nframes) v = X[f * D + c]; }
            else if (c == Dp) v = 1.f;
            else if (c == Dp + 1) v = kConstEps;
            xw[idx] = v;
        }
    }
}

// ---------------------------------------------------------------------------
// K1 on the bf16 pipes: one wave owns 16 MT frames x 16 NT components (a chunk of
// the K components: blockIdx via xcd_block when there are several).  The large
// form (MT = 4, NT = 16: 64 frames x 256 components, 256 accumulators, one wave per
// SIMD) builds every A fragment once for 256 components: the fragment arithmetic
// (4 products + 4.5 VALU per element for the three-way split) is what the fp16
// two-piece kernel of round 2 was bound by, with half the MFMAs per element.
// A fragments are built one k-step ahead, in slices between the MFMA batches; B
// fragments (three 16-byte loads per tile from the packed image) one batch of BT
// tiles ahead.
// SQ = false: no "square" slabs in the table (full covariance), the per-product
// select between x_j^2 and x_a x_j drops out of the A-fragment arithmetic.
// ---------------------------------------------------------------------------
// IMG: the A fragments come from the caller's frame fragment image (frame_image_kernel
// further down: they depend on the frames only), 16-byte loads instead of the staging of
// the frames and the fragment arithmetic.
// BL: the B fragments of a k-step go through LDS ONCE PER WORKGROUP -- half a k-step (8 tiles,
// 24 KiB) at a time, copied global -> LDS by the DMA path into a ring of two buffers -- instead
// of once per wave from L2: every wave used to stream the chunk's whole packed image (1.4 MB at
// D = 40) for its 32 frames, 43 GB per launch of config 2 through the L2s, and the kernel ran
// at 1.69 GHz where the accumulation holds 1.91 (r05_pmc.json).  Two synchronisation points per
// k-step (in front of the batches whose look-ahead reads cross into the other buffer): wait
// for the own DMA, barrier, refill the buffer that has just been read out.  A wave that waits
// at the barrier leaves the matrix pipe to the wave of the other workgroup on its SIMD.
template <int NT, int MT, int GQ, bool PACKED, bool SQ, bool LNO, bool IMG = false, bool BL = false>
__global__ __launch_bounds__(kThreads, MT * NT <= 32 ? 2 : 1) void llhx_kernel(
    int64_t nframes, int D, int K, int S, int G, int gl, int jw, int nk,
    const float* __restrict__ X, const u4* __restrict__ Pall, const int* __restrict__ tab,
    float* __restrict__ resps, float* __restrict__ log_norm, double* __restrict__ llh_sum,
    float* __restrict__ xt_out, int xt_floats, int nku, int cg, const float* __restrict__ c0,
    const u4* __restrict__ img = nullptr) {
    static_assert(!IMG || (MT == 2 && !PACKED && LNO), "the image holds 32-frame tiles");
    static_assert(!BL || (NT == 16 && MT == 2 && !IMG), "half k-steps of 8 tiles, hipcc-scheduled form");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int LD = ld16_of(D);                                // 16-byte aligned rows
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    constexpr int FW = 16 * MT, NW = kThreads / 64;
    constexpr bool PIN = MT * NT > 32;                        // 256 accumulators: see mfma_bf16_pinned
    float* xw = reinterpret_cast<float*>(smem) + wave * (FW * LD);
    int* tabs = reinterpret_cast<int*>(reinterpret_cast<float*>(smem) + NW * FW * LD);
    // (BL) the ring of two half-k-step buffers behind the slab table, 1 KiB aligned
    constexpr int kHalfBytes = 8 * kBlockU4 * 16;             // 8 tiles x 3 KiB
    char* ring = smem + (((size_t)NW * FW * LD * sizeof(float) + (size_t)(nk + 1) * 8 * sizeof(int)
                          + 1023) & ~(size_t)1023);
    int64_t bx = blockIdx.x;
    int by = 0;
    {
        const int nch = (K + 16 * NT - 1) / (16 * NT);
        constexpr int FBK = FW * NW;
        if (nch > 1 && !xcd_block((nframes + FBK - 1) / FBK, nch, cg, bx, by)) return;
    }
    const int64_t fb = (bx * NW + wave) * FW;
    for (int idx = tid; idx < (nk + 1) * 8; idx += kThreads) tabs[idx] = tab[idx];
    if (!IMG) stage_rows<FW>(X, fb, nframes, D, LD, lane, xw);
    __syncthreads();
    if (IMG && fb >= nframes) return;
    {
        // The wave's frames are (a part of) one 64-frame tile of the accumulation kernel:
        // leave them behind transposed, [D + 2][68] (rows D, D + 1 = 1, 0; see
        // xt_image_kernel, which this replaces -- one pass over the frames less).  A wave
        // of 32 frames writes its 32 columns; the one that holds the tile's last frames
        // also writes the 4 columns of padding.
        static_assert(64 % FW == 0, "whole waves per 64-frame tile");
        if (xt_out && by == 0 && fb < (nframes + 63) / 64 * 64) {
            float* img = xt_out + (fb / 64) * (size_t)xt_floats;
            const int c0 = (int)(fb % 64);
            const int ncol4 = FW / 4 + (c0 + FW == 64 ? 1 : 0);
            for (int e4 = lane; e4 < (D + 2) * ncol4; e4 += 64) {
                const int row = e4 / ncol4, c = 4 * (e4 - row * ncol4);      // column c0 + c
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (row < D) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = c + j < FW ? xw[(c + j) * LD + row] : 0.f;
                } else if (row == D) {
                    v = f32x4{1.f, 1.f, 1.f, 1.f};
                }
                __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(img + row * 68 + c0 + c));
            }
        }
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int c = 0; c < NT; ++c) acc[m][c] = f32x4{0, 0, 0, 0};

    const float* xrow[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) xrow[m] = xw + (m * 16 + i) * LD;

    const int kbase = by * (16 * NT);
    // the B stream of this chunk is one linear sequence of (k-step, tile) blocks of
    // 3 KiB = 192 u4 (three planes of 64 lanes x 16 B)
    const u4* Pl = Pall + (size_t)by * nk * NT * kBlockU4 + lane;
    const int* tl = tabs + 2 * g;

    // A fragments as 32-bit words (two bf16 each): word w of piece q of tile m holds
    // the entries 2w, 2w+1 of the lane's 8-deep k-block (w < 2: first slab).
    struct AFrag { u4 w[NP][MT]; };
    // one slab (half a k-block) of tile m of k-step s: LDS reads, 4 products, the
    // three-way split -> words 2h, 2h+1 of every piece
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
            unsigned w3[3];
            split3(p[2 * e], p[2 * e + 1], w3);
#pragma unroll
            for (int q = 0; q < NP; ++q) f.w[q][m][2 * h + e] = w3[q];
        }
    };
    // B fragments: BT column tiles per batch, loaded one batch ahead of their MFMAs
    // (sched_group_barrier pins the loads above the MFMA block: hipcc otherwise sinks
    // every load next to its first use)
    constexpr int BT = 2, NBATCH = NT / BT;
    static_assert(NT % BT == 0 && NBATCH % 2 == 0, "B batches alternate between two buffers");
    struct BFrag { u4 p[BT][NP]; };
    auto load_b = [&](int64_t blk, BFrag& b) {               // blocks blk .. blk + BT - 1
#pragma unroll
        for (int c = 0; c < BT; ++c)
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                if constexpr (BL) {
                    // block blk + c sits in buffer ((blk + c) / 8) & 1 at position (blk + c) & 7
                    const int bb = (int)(blk + c);
                    b.p[c][q] = *reinterpret_cast<const u4*>(
                        ring + ((bb >> 3) & 1) * kHalfBytes + (bb & 7) * (kBlockU4 * 16) + q * 1024 +
                        lane * 16);
                } else {
                    b.p[c][q] = Pl[(size_t)(blk + c) * kBlockU4 + 64 * q];
                }
            }
    };
    // (BL) DMA of half k-step `hs` (blocks 8 hs .. 8 hs + 7 of the chunk's image) into buffer
    // hs & 1: 24 pieces of 1 KiB, six per wave
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const char* bsrc = reinterpret_cast<const char*>(Pall + (size_t)by * nk * NT * kBlockU4) +
                       wave * 1024 + lane * 16;
    auto stage_half = [&](int hs) {
        const char* src = bsrc + (size_t)hs * kHalfBytes;
        char* dst = ring + (hs & 1) * kHalfBytes + wave * 1024;
#pragma unroll
        for (int n = 0; n < kHalfBytes / 1024 / NW; ++n)
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const u4*>(src + n * (NW * 1024)),
                                             (lds_ptr)(dst + n * (NW * 1024)), 16, 0, 0);
    };
    auto publish = [&]() {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    auto batch = [&](int s, int bi, const AFrag& cur, AFrag& nxt, const BFrag& b, BFrag& bn) {
        if constexpr (PIN) {
            // Hand-placed stream: one wave per SIMD issues in order, so whatever is not
            // interleaved with the MFMAs runs while the matrix pipe idles (the block form
            // below measured 52 % MFMA busy).  48 pinned MFMAs, each followed by at most
            // one filler: the 6 loads of the next B batch, then ONE half A fragment of the
            // next k-step -- table entry, 2 LDS reads, 4 products, 2 x 7 steps of the
            // three-way split -- and a scheduling fence, so that hipcc keeps the order.
            static_assert(!PIN || (MT * 2 == NBATCH && 6 * BT * MT == 48), "one half fragment per batch");
            const int hm = bi % MT, hh = bi / MT;             // the half built in this batch
            const int64_t blk = (int64_t)s * NT + (bi + 1) * BT;
            int t = 0;
            f32x4 bb = {0.f, 0.f, 0.f, 0.f}, p = {0.f, 0.f, 0.f, 0.f};
            float xx = 0.f;
            Split3Steps st[2];
#pragma unroll
            for (int n = 0; n < 48; ++n) {
                const int pr = n / (BT * MT), c = (n / MT) % BT, m = n % MT;
                mfma_bf16_pinned<false>(acc[m][bi * BT + c], cur.w[kProdA[pr]][m],
                                        b.p[c][kProdB[pr]]);
                if (BEER_K1_ABL >= 1 && n >= NP * BT) {
                } else if (BEER_K1_ABL >= 2) {
                } else if (n < NP * BT) {
#ifdef BEER_K1_FAKEB
                    // (timing experiment: every load hits the same 12 KB -- wrong results)
                    bn.p[n / NP][n % NP] = Pl[(size_t)((blk + n / NP) & 3) * kBlockU4 + 64 * (n % NP)];
#else
                    bn.p[n / NP][n % NP] = Pl[(size_t)(blk + n / NP) * kBlockU4 + 64 * (n % NP)];
#endif
                } else if (n == 8) {
                    t = tl[8 * (s + 1) + hh];                 // (the table is padded by one k-step)
                } else if (n == 12) {
                    bb = *reinterpret_cast<const f32x4*>(xrow[hm] + ((t >> 8) & 0xff));
                    xx = xrow[hm][t & 0xff];
                } else if (n >= 18 && n < 22) {
                    const bool sq = SQ && (t >> 16) != 0;
                    float v = bb[n - 18] * (sq ? bb[n - 18] : xx);
                    pin(v);
                    p[n - 18] = v;
                } else if (n >= 22 && n < 29) {
                    split3_step(n - 22, p[0], p[1], st[0]);
                } else if (n >= 29 && n < 36) {
                    split3_step(n - 29, p[2], p[3], st[1]);
                }
                if (BEER_K1_FENCE > 0 && n % BEER_K1_FENCE == BEER_K1_FENCE - 1)
                    __builtin_amdgcn_sched_barrier(0);
            }
            if (BEER_K1_ABL >= 1) {
#pragma unroll
                for (int q = 0; q < NP; ++q) nxt.w[q][hm] = cur.w[q][hm];
                if (BEER_K1_ABL >= 2) bn = b;
                return;
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                nxt.w[0][hm][2 * hh + e] = st[e].w0;
                nxt.w[1][hm][2 * hh + e] = st[e].w1;
                nxt.w[2][hm][2 * hh + e] = st[e].w2;
            }
            return;
        }
        if constexpr (BL) {
            // in front of the batches whose look-ahead read crosses into the other buffer:
            // batch 3 (reads tile 8 of this k-step; buffer 0 has been read out: refill it with
            // the first half of the next k-step), batch 7 (reads tile 0 of the next k-step;
            // buffer 1 is free for its second half)
            if (bi == NBATCH / 2 - 1 || bi == NBATCH - 1) {
                publish();
                const int hs = 2 * (s + 1) + (bi == NBATCH - 1 ? 1 : 0);
                if (hs < 2 * nku) stage_half(hs);
            }
        }
        // P is padded by one batch (bi + 1 = NBATCH: first batch of the next k-step)
        load_b((int64_t)s * NT + (bi + 1) * BT, bn);
        // slices of the next A: MT * 2 halves over the NBATCH batches
        // (half-major: the halves of one batch share the table entry)
        if constexpr (!IMG) {
#pragma unroll
            for (int hh = bi * MT * 2 / NBATCH; hh < (bi + 1) * MT * 2 / NBATCH; ++hh)
                make_half(s + 1, hh % MT, hh / MT, nxt);     // the table is padded by one k-step
        }
#pragma unroll
        for (int pr = 0; pr < 6; ++pr)
#pragma unroll
            for (int c = 0; c < BT; ++c)
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    acc[m][bi * BT + c] = mfma_bf16(cur.w[kProdA[pr]][m], b.p[c][kProdB[pr]],
                                                    acc[m][bi * BT + c]);
        if constexpr (BL)
            __builtin_amdgcn_sched_group_barrier(0x100, NP * BT, 0);     // DS reads (B fragments)
        else
            __builtin_amdgcn_sched_group_barrier(0x020, NP * BT, 0);     // VMEM reads
        if constexpr (!IMG)
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * ((MT * 2 + NBATCH - 1) / NBATCH), 0);   // DS reads
        __builtin_amdgcn_sched_group_barrier(0x008, 6 * MT * BT, 0);     // MFMA
    };
    // (IMG) the image tile of this wave: [k-step][piece][frame tile][64 lanes] u4, then the
    // statistics' fragments, which this kernel does not use
    const u4* ti = nullptr;
    if constexpr (IMG)
        ti = img + (fb / FW) * (int64_t)frame_image_tile_u4(nku, img_nqt(nku)) + lane;
    auto load_a = [&](int s, AFrag& f) {
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int m = 0; m < MT; ++m) f.w[q][m] = ti[((s * NP + q) * MT + m) * 64];
    };
    auto kstep = [&](int s, const AFrag& cur, AFrag& nxt, BFrag& b0, BFrag& b1) {
        if constexpr (IMG) {
            if (s + 1 < nku) load_a(s + 1, nxt);
        }
#pragma unroll
        for (int bi = 0; bi < NBATCH; bi += 2) {
            batch(s, bi, cur, nxt, b0, b1);
            batch(s, bi + 1, cur, nxt, b1, b0);
        }
    };
    AFrag f0, f1;
    BFrag b0, b1;
    if constexpr (IMG) {
        load_a(0, f0);
    } else {
#pragma unroll
        for (int hh = 0; hh < MT * 2; ++hh) make_half(0, hh % MT, hh / MT, f0);
    }
    if constexpr (BL) {
        stage_half(0);
        if (nku > 0) stage_half(1);
        publish();
    }
    load_b(0, b0);
    // (the image is padded to an even number of k-steps; only those that hold slabs run)
    for (int s = 0; s < nku; s += 2) {
        kstep(s, f0, f1, b0, b1);
        if (s + 1 < nku) kstep(s + 1, f1, f0, b0, b1);
    }
    if constexpr (PIN) mfma_drain();
    if (PIN && BEER_K1_ABL >= 3) {
        if (acc[0][0][0] == 1.2345f) log_norm[0] = 1.f;
        return;
    }
    if constexpr (LNO) {
        // (jw == 4 by construction of the dispatch; gl = lanes per group, uniform; 0: the
        // image is lane-major, a group is G <= 16 registers of one lane)
        if constexpr (NT == 16 && GQ == 1) {
            if (gl == 0) {
                if (G == 4)
                    lognorm_epilogue_lane_major<NT, MT, 4>(acc, fb, nframes, kbase, S, i, g, lane, log_norm, llh_sum, c0[0]);
                else if (G == 8)
                    lognorm_epilogue_lane_major<NT, MT, 8>(acc, fb, nframes, kbase, S, i, g, lane, log_norm, llh_sum, c0[0]);
                else
                    lognorm_epilogue_lane_major<NT, MT, 16>(acc, fb, nframes, kbase, S, i, g, lane, log_norm, llh_sum, c0[0]);
                return;
            }
        }
#define BEER_LN(GL_) lognorm_epilogue<NT, MT, GQ, GL_>(acc, fb, nframes, kbase, S, G, i, g, lane, \
                                                       log_norm, llh_sum, c0[0])
        switch (gl) {
            case 1: BEER_LN(1); break;
            case 2: BEER_LN(2); break;
            case 4: BEER_LN(4); break;
            case 8: BEER_LN(8); break;
            default: BEER_LN(16); break;
        }
#undef BEER_LN
    } else {
        softmax_epilogue<float, NT, MT, GQ, PACKED, LNO>(acc, fb, nframes, kbase, K, S, G, gl, jw,
                                                         i, g, lane, resps, log_norm, llh_sum,
                                                         c0[0]);
    }
}

// Two workgroups of the LDS-staged form (BL) fit a CU: frame tiles + slab table + the ring
inline bool k1_lds_fits(int D, int nk) {
    size_t lds = (size_t)32 * (kThreads / 64) * ld16_of(D) * sizeof(float) + (size_t)(nk + 1) * 8 * sizeof(int);
    lds = ((lds + 1023) & ~(size_t)1023) + 2 * (size_t)(8 * kBlockU4 * 16);
    return 2 * lds <= (size_t)beer::kMaxDynLds;
}

// covariance type of the E-step being launched (the launch helpers below take the
// shape, not the type; SQ = false kernels are full covariance by construction)
thread_local int g_cov_of_launch = BEER_FULL;

template <int NT, int MT, int GQ, bool PACKED = false, bool SQ = true, bool LNO = false,
          bool IMG = false, bool BL = false>
int launch_llhx(int64_t nframes, int D, int K, int S, int G, int gl, int jw, int nchunks, int nk,
                const float* X, const void* P, const int* tab, const float* c0, float* resps,
                float* log_norm, double* llh_sum, hipStream_t s, float* xt_out = nullptr,
                int xt_floats = 0, const void* img = nullptr) {
    const int LD = ld16_of(D);
    // k-steps that hold slabs: the slab count is the table's (full: SQ = false)
    const int nku = (nslab_of(SQ ? g_cov_of_launch : BEER_FULL, D) + 7) / 8;
    constexpr int FB = 16 * MT * (kThreads / 64);
    size_t lds = (size_t)FB * LD * sizeof(float) + (size_t)(nk + 1) * 8 * sizeof(int);
    if (BL) lds = ((lds + 1023) & ~(size_t)1023) + 2 * (size_t)(8 * kBlockU4 * 16);
    const int64_t blocks = (nframes + FB - 1) / FB;
    if (nchunks != (K + 16 * NT - 1) / (16 * NT)) return BEER_EINVAL;
    const int cg = xcd_chunk_group(nchunks, (size_t)nku * NT * kBlockU4 * 16);
    (void)hipFuncSetAttribute(
        reinterpret_cast<const void*>(llhx_kernel<NT, MT, GQ, PACKED, SQ, LNO, IMG, BL>),
        hipFuncAttributeMaxDynamicSharedMemorySize, beer::kMaxDynLds);
    hipLaunchKernelGGL((llhx_kernel<NT, MT, GQ, PACKED, SQ, LNO, IMG, BL>),
                       dim3(nchunks > 1 ? xcd_grid(blocks, nchunks, cg) : (unsigned)blocks),
                       dim3(kThreads), lds, s, nframes, D, K, S, G, gl, jw, nk, X,
                       reinterpret_cast<const u4*>(P), tab, resps, log_norm, llh_sum, xt_out,
                       xt_floats, nku, cg, c0, reinterpret_cast<const u4*>(img));
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

// ---------------------------------------------------------------------------
// K2 on the bf16 pipes: S[k, q] += sum_t r[t,k] * PHI_q(x_t) with frames as the
// contraction index (32 per MFMA), both operands as ready-made LDS images -- R as
// K1 packed it (estep_tiles.h: three planes per 64 x 128 tile), X as K1 /
// xt_image_kernel transposed it -- copied global -> LDS by the DMA path
// (global_load_lds_dwordx4: the images are lane-linear by construction), tile t + 1
// in flight while tile t is multiplied, one barrier per tile.
// Workgroup = 4 waves, one per SIMD; a wave owns 128 components x 8 statistic
// tiles (256 accumulators): A fragments = three ds_read_b128 per component tile,
// loaded once per k-step and reused by the 8 statistic tiles; B fragments =
// products of two X^T rows over the lane's 8 frames, split on the fly, reused by
// the 8 component tiles.  A workgroup sums at most BEER_OPT_AX_MAXFRAMES frames in fp32 --
// one rounding per 32-frame MFMA -- and adds its partial sums to the fp64 image
// with atomics.
// SR (mixture sets): the tiles hold the responsibilities WITHIN each state's
// mixture (what the E-step knows); the state posteriors of the forward-backward
// pass that ran in between arrive as transposed tiles Gt [64-frame tile][state][64
// frames] and are multiplied in while the tile sits in LDS, each element once per
// workgroup: (p0 + p1 + p2) * gamma in fp32, split again (in place, between the
// arrival of the tile and its first use).  lgG = log2 of the components per state
// (8 .. 128: the states of a 128-component block fit 4 KiB).
// ---------------------------------------------------------------------------
constexpr int kAxFT = 64;             // frames per LDS tile (2 k-steps)
constexpr int kAxXS = kAxFT + 4;      // X^T row stride (floats), 16-byte aligned
// Frames per workgroup = length of an fp32 accumulation chain (one MFMA per 32
// frames).  The bf16 MFMA aligns its 32 products to the largest addend -- the
// accumulator, once it has grown -- and TRUNCATES each of them toward zero at about
// ulp(C) / 32 (measured: tools/probes/mfma_round.hip): a chain over products of one
// sign (N_k = sum r, sum r x^2) loses ~ulp(C) / 2 per instruction, i.e. n * 2e-8 of
// the sum after n accumulations -- 1e-5 for the 16384-frame chains of round 2, which
// is what that round's "parameter rounding bias" really was.  Measured on the counts of
// a 1 M-frame, 256-component mixture (tools/probes/chain_len.py): -1.1e-7 with chains of
// 1024 frames, -3.3e-7 with 4096, -1.5e-6 with 16384.  4096 it is; the partial sums of the
// workgroups are added in fp64 (the flush costs 0.1 ms per 1 M frames and 2048 frames of
// chain at K = 256).
// (the chain length: BEER_OPT_AX_MAXFRAMES, default beer::kAxMaxFramesDefault = 4096)
constexpr int kPiece = 4096;          // granule of the X^T image (bytes)
constexpr int kAxMC = 8, kAxNQ = 8, kAxWaves = 4;

inline int xt_rows(int D) { return D + 2; }                              // + ones, zeros
inline int xt_pieces(int D) { return (xt_rows(D) * kAxXS * 4 + kPiece - 1) / kPiece; }

// X [T, D] -> per 64-frame tile the image [D + 2][kAxXS] of the frames (rows D,
// D + 1 = the constants 1, 0; frames past T = 0), padded to whole pieces
__global__ __launch_bounds__(256) void xt_image_kernel(int64_t nframes, int D, int NX,
                                                       const float* __restrict__ X,
                                                       float* __restrict__ Xt) {
    constexpr int LD = kMaxDimX + 1;              // odd: the transposed reads spread over the banks
    __shared__ float tile[kAxFT * LD];
    const int64_t tau = blockIdx.x, t0 = tau * kAxFT;
    const int rows = (int)(nframes - t0 < kAxFT ? nframes - t0 : kAxFT);
    for (int e = threadIdx.x; e < rows * D; e += 256) {
        const int f = e / D, d = e - f * D;
        tile[f * LD + d] = X[t0 * D + e];
    }
    __syncthreads();
    float* out = Xt + tau * ((size_t)NX * (kPiece / 4));
    for (int e = threadIdx.x; e < NX * (kPiece / 4); e += 256) {
        const int row = e / kAxXS, col = e - row * kAxXS;
        float v = 0.f;
        if (row < D) v = col < rows ? tile[col * LD + row] : 0.f;
        else if (row == D) v = 1.f;
        __builtin_nontemporal_store(v, out + e);
    }
}

template <bool SR>
__global__ __launch_bounds__(64 * kAxWaves, 1) void accx_kernel(
    int64_t nframes, int D, int K, int nslab, int NX, const float* __restrict__ Xt,
    const unsigned* __restrict__ Rimg, const int* __restrict__ tab,
    int64_t frames_per_block, double* __restrict__ Sp, int gx, int gy, int gz,
    const float* __restrict__ Gt, int lgG, int sx) {
    constexpr int MC = kAxMC, NQ = kAxNQ, WAVES = kAxWaves, NB = 2;
    static_assert(16 * MC == kPackedComps && kAxFT == kPackedFrames, "the packed image is this kernel's LDS tile");
    constexpr int plane = kPackedPlaneWords * 4;                  // bytes of one piece plane
    // NX = 4 KiB pieces of the transposed frame tile (1 .. 9: D <= 128).  Two buffers of
    // [X^T | three planes of R] -- or, `sx` (D > 112: two of them do not fit a CU's 160 KiB),
    // ONE X^T tile in front of two buffers of R planes; the next tile's X^T is then fetched
    // only when every wave is done with this one (an exposed DMA round trip per tile: the
    // price of the shapes beyond 112 dimensions, still an order of magnitude ahead of the
    // generic kernels).
    const int r_off = NX * kPiece, buf_bytes = r_off + NP * plane;
    const int xstride = sx ? 0 : buf_bytes;                       // X^T of buffer b at b * xstride
    const int rstride = sx ? NP * plane : buf_bytes;              // R planes of buffer b at r_off + b * rstride
    const int g_base = sx ? r_off + NB * NP * plane : NB * buf_bytes;   // SR: NB x 4 KiB of gamma^T
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int D4 = d4_of(D), Dp = 4 * D4, nq = nslab * 4;
    // XCD-aware mapping: workgroups are dealt round-robin to the 8 XCDs, each with
    // its own L2.  The gx statistic blocks that read the same R tile get ids
    // congruent mod 8, i.e. the same XCD back to back: one of them misses in L2,
    // the others hit.
    const int id = blockIdx.x, xcd = id & 7, slot0 = id >> 3;
    const int bx = slot0 % gx;
    const int yz = (slot0 / gx) * 8 + xcd;
    if (yz >= gy * gz) return;
    const int by = yz % gy, bz = yz / gy;
    const int tile0 = (bx * WAVES + wave) * NQ;
    const int kc0 = by * (16 * MC);
    const int64_t tb = (int64_t)bz * frames_per_block;
    const int64_t te = tb + frames_per_block < nframes ? tb + frames_per_block : nframes;
    const int ntiles = (int)((te - tb + kAxFT - 1) / kAxFT);
    const int64_t tau0 = tb / kAxFT;
    const int nblk = (K + 16 * MC - 1) / (16 * MC);
    const int gbytes = SR ? ((16 * MC) >> lgG) * kAxFT * 4 : 0;   // gamma^T of one block-tile
    const int gkb = (gbytes + 1023) >> 10;

    auto factors = [&](int uu, int& a, int& b) {
        const int col = 16 * (tile0 + uu) + i, slab = col >> 2;
        a = b = Dp + 2;
        if (slab < nslab) {
            const int t = tab[slab];
            b = ((t >> 8) & 0xff) + (col & 3);
            a = (t >> 16) ? b : (t & 0xff);
        }
    };
    // byte offsets of the lane's 8 frames (k-step 0) in the two X^T rows of its
    // statistic column, per tile of the wave (row D = ones, row D + 1 = zeros)
    int xa_off[NQ], xb_off[NQ];
#pragma unroll
    for (int uu = 0; uu < NQ; ++uu) {
        int a, b;
        factors(uu, a, b);
        xa_off[uu] = ((a < D ? a : (a == Dp ? D : D + 1)) * kAxXS + 8 * g) * 4;
        xb_off[uu] = ((b < D ? b : (b == Dp ? D : D + 1)) * kAxXS + 8 * g) * 4;
    }
    // A fragments: row i of component tile c, chunk (4 ks + g) ^ (i & 7)
    int a_off[kAxFT / 32];
#pragma unroll
    for (int ks = 0; ks < kAxFT / 32; ++ks)
        a_off[ks] = r_off + (i * kPackedFrames + (((4 * ks + g) ^ (i & 7)) << 3)) * 2;   // (+ b * rstride)

    f32x4 acc[MC][NQ];
#pragma unroll
    for (int c = 0; c < MC; ++c)
#pragma unroll
        for (int uu = 0; uu < NQ; ++uu) acc[c][uu] = f32x4{0, 0, 0, 0};

    // DMA of tile `tile` into buffer `buf`: every wave-instruction moves 1 KiB; wave w
    // takes the blocks w, w + 4, ... of the X image (4 NX blocks), of the three planes of
    // the R image of block `by` (48 blocks) and, with SR, of gamma^T (<= 4 blocks).  The
    // regions are whole multiples of the 4 waves, so which region a piece belongs to is
    // known at compile time: no branches, one per-lane base address per region.  (A
    // branch chain per piece cost scalar registers; their spill reloads came with
    // s_waitcnt vmcnt(0), which also waits for the DMA issued just before -- 16 serialised
    // memory round trips per tile, 46 k cycles where the MFMAs need 12 k.)
    static_assert((kPiece / 1024) % WAVES == 0 && (NP * plane / 1024) % WAVES == 0,
                  "regions are whole rounds of the waves");
    const char* xsrc = reinterpret_cast<const char*>(Xt) + wave * 1024 + lane * 16;
    const char* rsrc = reinterpret_cast<const char*>(Rimg) + wave * 1024 + lane * 16;
    const char* gsrc = SR ? reinterpret_cast<const char*>(Gt) + wave * 1024 + lane * 16 : nullptr;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    auto stage_x = [&](int tile, int buf) {
        const char* xs = xsrc + (tau0 + tile) * (size_t)(NX * kPiece);
        char* dst = smem + buf * xstride + wave * 1024;
#pragma unroll 1
        for (int n = 0; n < NX * (kPiece / 1024) / WAVES; ++n)
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const u4*>(xs + n * (WAVES * 1024)),
                                             (lds_ptr)(dst + n * (WAVES * 1024)), 16, 0, 0);
    };
    auto stage_r = [&](int tile, int buf) {
        const int64_t tau = tau0 + tile;
        const char* rs = rsrc + (tau * nblk + by) * (size_t)(NP * plane);
        char* dst = smem + r_off + buf * rstride + wave * 1024;
#pragma unroll
        for (int n = 0; n < NP * plane / 1024 / WAVES; ++n)
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const u4*>(rs + n * (WAVES * 1024)),
                                             (lds_ptr)(dst + n * (WAVES * 1024)), 16, 0, 0);
        if (SR && wave < gkb)
            __builtin_amdgcn_global_load_lds(
                reinterpret_cast<const u4*>(gsrc + (tau * nblk + by) * (size_t)gbytes),
                (lds_ptr)(smem + g_base + buf * 4096 + wave * 1024), 16, 0, 0);
    };
    auto stage = [&](int tile, int buf) {
        if (!sx) stage_x(tile, buf);
        stage_r(tile, buf);
    };
    // SR: this thread's share of the R image in `buf`: 4 chunks of 8 frames (row c,
    // chunk position pos holds frames 8 (pos ^ (c & 7)) ..), all three planes, times
    // the state's gamma
    auto fold = [&](int buf) {
        if (BEER_K2_ABL == 4) return;                       // (timing experiment: no fold)
#pragma unroll 1
        for (int n = 0; n < 16 * MC * 8 / (64 * WAVES); ++n) {
            const int p = tid + 64 * WAVES * n, c = p >> 3, pos = p & 7, ch = pos ^ (c & 7);
            char* ph = smem + buf * rstride + r_off + c * (kPackedFrames * 2) + pos * 16;
            const u4 w0 = *reinterpret_cast<const u4*>(ph);
            const u4 w1 = *reinterpret_cast<const u4*>(ph + plane);
            const u4 w2 = *reinterpret_cast<const u4*>(ph + 2 * plane);
            const float* gm = reinterpret_cast<const float*>(smem + g_base + buf * 4096) +
                              ((c >> lgG) * kAxFT + 8 * ch);
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(gm);
            const f32x4 g1 = *reinterpret_cast<const f32x4*>(gm + 4);
            u4 o0, o1, o2;
#pragma unroll
            for (int wd = 0; wd < 4; ++wd) {
                // p1 + p2 is exact (16 bits), + p0 restores the float32 value
                const float lo = (__builtin_bit_cast(float, w1[wd] << 16) +
                                  __builtin_bit_cast(float, w2[wd] << 16)) +
                                 __builtin_bit_cast(float, w0[wd] << 16);
                const float hi = (__builtin_bit_cast(float, w1[wd] & 0xffff0000u) +
                                  __builtin_bit_cast(float, w2[wd] & 0xffff0000u)) +
                                 __builtin_bit_cast(float, w0[wd] & 0xffff0000u);
                const float ga = wd < 2 ? g0[2 * wd] : g1[2 * wd - 4];
                const float gb = wd < 2 ? g0[2 * wd + 1] : g1[2 * wd - 3];
                unsigned w3[3];
                split3(lo * ga, hi * gb, w3);
                o0[wd] = w3[0]; o1[wd] = w3[1]; o2[wd] = w3[2];
            }
            *reinterpret_cast<u4*>(ph) = o0;
            *reinterpret_cast<u4*>(ph + plane) = o1;
            *reinterpret_cast<u4*>(ph + 2 * plane) = o2;
        }
    };
    if (ntiles > 0) {
        stage_x(0, 0);
        stage_r(0, 0);
    }
    __syncthreads();
    if (SR && ntiles > 0) {
        fold(0);
        __syncthreads();
    }
    const bool active = tile0 * 16 < nq;
    // A k-step (32 frames): the A fragments of the 8 component tiles are loaded at its
    // start -- all three pieces at once, into registers that are dead by then; the
    // products run piece 0 of A first, so only those 8 reads are waited for --, the B
    // fragment of statistic tile uu + 1 is built while tile uu is multiplied (the one of
    // the next k-step's tile 0 during tile 7).
    u4 af[MC][NP], bf[2][NP];
    // the six products with the pieces of A in ascending order: (A, B) =
    constexpr int kPA[6] = {0, 0, 0, 1, 1, 2}, kPB[6] = {0, 1, 2, 0, 1, 0};
    auto a_ptr = [&](int b, int ks, int c) {
        return smem + a_off[ks] + (b * rstride + c * 16 * kPackedFrames * 2);
    };
    auto load_a = [&](int b, int ks) {
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int c = 0; c < MC; ++c)
                af[c][q] = *reinterpret_cast<const u4*>(a_ptr(b, ks, c) + q * plane);
    };
    auto gen_b = [&](int b, int ks, int uu, u4 (&out)[NP]) {
        const char* pa = smem + xa_off[uu] + (b * xstride + 128 * ks);
        const char* pb = smem + xb_off[uu] + (b * xstride + 128 * ks);
        const f32x4 xa0 = *reinterpret_cast<const f32x4*>(pa);
        const f32x4 xa1 = *reinterpret_cast<const f32x4*>(pa + 16);
        const f32x4 xb0 = *reinterpret_cast<const f32x4*>(pb);
        const f32x4 xb1 = *reinterpret_cast<const f32x4*>(pb + 16);
        const f32x4 p0 = xa0 * xb0, p1 = xa1 * xb1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            unsigned w3[3];
            split3(e < 2 ? p0[2 * e] : p1[2 * e - 4], e < 2 ? p0[2 * e + 1] : p1[2 * e - 3], w3);
#pragma unroll
            for (int q = 0; q < NP; ++q) out[q][e] = w3[q];
        }
    };
    if (active && ntiles > 0) gen_b(0, 0, 0, bf[0]);
    // One tile out of buffer `buf` (a constant once inlined): 16 steps of 48 MFMAs, then
    // ONE barrier (two with SR): every wave is done reading `buf`, and the DMA of tile
    // + 1 into the other buffer, issued at the start of this tile, has landed (the
    // barrier's vmcnt(0)).
    auto iteration = [&](int tile, int buf) __attribute__((always_inline)) {
        const int next = buf ^ 1;
        if (tile + 1 < ntiles) stage(tile + 1, next);
        const bool fold_next = SR && tile + 1 < ntiles;
        if (active) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (BEER_K2_ABL < 2 || (tile == 0 && ks == 0)) load_a(buf, ks);
#pragma unroll
                for (int uu = 0; uu < NQ; ++uu) {
                    const int cur = uu & 1;
                    // The B fragment of the next statistic tile (of tile 0 of the next
                    // k-step; the one of the next TILE comes from the other buffer, after
                    // the barrier), built behind this tile's 48 MFMAs, one filler each, every
                    // result pinned where it is computed (see K1): 4 LDS reads, 8 products,
                    // 4 x 7 steps of the three-way split.
                    const bool build = BEER_K2_ABL < 1 && (uu + 1 < NQ || ks == 0);
                    const int nks = uu + 1 < NQ ? ks : 1, nuu = uu + 1 < NQ ? uu + 1 : 0;
                    const char* pa = smem + xa_off[nuu] + (buf * xstride + 128 * nks);
                    const char* pb = smem + xb_off[nuu] + (buf * xstride + 128 * nks);
                    f32x4 xa[2], xb[2];
                    float pp[8];
                    Split3Steps st[4];
#pragma unroll
                    for (int n = 0; n < 6 * MC; ++n) {
                        const int pr = n / MC, c = n % MC;
                        mfma_bf16_pinned<false>(acc[c][uu], af[c][kPA[pr]], bf[cur][kPB[pr]]);
                        if (!build) continue;
                        if (n == 1) xa[0] = *reinterpret_cast<const f32x4*>(pa);
                        else if (n == 2) xa[1] = *reinterpret_cast<const f32x4*>(pa + 16);
                        else if (n == 3) xb[0] = *reinterpret_cast<const f32x4*>(pb);
                        else if (n == 4) xb[1] = *reinterpret_cast<const f32x4*>(pb + 16);
                        else if (n >= 12 && n < 20) {
                            const int e = n - 12;
                            if (BEER_ASM_STEPS) {
                                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(pp[e])
                                             : "v"(xa[e >> 2][e & 3]), "v"(xb[e >> 2][e & 3]));
                            } else {
                                pp[e] = xa[e >> 2][e & 3] * xb[e >> 2][e & 3];
                                pin(pp[e]);
                            }
                        } else if (n >= 20) {
                            // two pairs at a time, their seven steps alternating: a step and the
                            // one that consumes its result are two MFMAs apart (back to back --
                            // one MFMA apart -- hipcc pads the dependence with an s_nop: 612 of
                            // them per tile pair)
#ifndef BEER_K2_CHAINS
#define BEER_K2_CHAINS 2
#endif
                            constexpr int NCH = BEER_K2_CHAINS;
                            const int m = n - 20, grp = m / (7 * NCH), mm = m % (7 * NCH);
                            const int e = NCH * grp + mm % NCH;
                            split3_step(mm / NCH, pp[2 * e], pp[2 * e + 1], st[e]);
                        }
                    }
                    if (build) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            bf[cur ^ 1][0][e] = st[e].w0;
                            bf[cur ^ 1][1][e] = st[e].w1;
                            bf[cur ^ 1][2][e] = st[e].w2;
                        }
                    } else if (BEER_K2_ABL >= 1) {
#pragma unroll
                        for (int q = 0; q < NP; ++q) bf[cur ^ 1][q] = bf[cur][q];
                    }
                }
            }
        }
        __syncthreads();
        if (sx && tile + 1 < ntiles) {
            stage_x(tile + 1, 0);              // every wave is done with this tile's X^T
            __syncthreads();
        }
        if (SR) {
            if (fold_next) fold(next);
            __syncthreads();
        }
        if (active && tile + 1 < ntiles) gen_b(next, 0, 0, bf[0]);
    };
    for (int tile = 0; tile < ntiles; tile += NB) {
        iteration(tile, 0);
        if (tile + 1 < ntiles) iteration(tile + 1, 1);
    }
    mfma_drain();
    if (BEER_K2_ABL >= 3) {
        if (acc[0][0][0] == 1.2345f) Sp[0] = 1.0;
        return;
    }
#pragma unroll
    for (int uu = 0; uu < NQ; ++uu) {
        const int q = (tile0 + uu) * 16 + i;
        if (q >= nq) continue;
#pragma unroll
        for (int c = 0; c < MC; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = kc0 + 16 * c + 4 * g + r;
                if (k < K) atomicAdd(Sp + (size_t)k * nq + q, (double)acc[c][uu][r]);
            }
    }
}

// State posteriors [T, S] -> transposed tiles Gt [tile of 64 frames][Spad states][64
// frames] (Spad = states of the padded component blocks; frames >= T and states >= S
// are 0): what accx_kernel<.., SR> copies to LDS next to a tile of responsibilities.
__global__ __launch_bounds__(256) void gt_image_kernel(int64_t nframes, int S, int Spad,
                                                       const float* __restrict__ sr,
                                                       float* __restrict__ Gt) {
    __shared__ float tile[64 * 65];
    const int64_t tau = blockIdx.x, t0 = tau * kAxFT;
    const int s0 = blockIdx.y * 64;
    for (int idx = threadIdx.x; idx < 64 * 64; idx += 256) {
        const int f = idx >> 6, st = idx & 63;
        tile[f * 65 + st] = (t0 + f < nframes && s0 + st < S) ? sr[(t0 + f) * S + s0 + st] : 0.f;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 64 * 64; idx += 256) {
        const int st = idx >> 6, f = idx & 63;
        if (s0 + st < Spad) Gt[(tau * Spad + s0 + st) * kAxFT + f] = tile[f * 65 + st];
    }
}

__device__ __forceinline__ float bf16_lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned w) {
    return __builtin_bit_cast(float, w & 0xffff0000u);
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
    const unsigned* w = Rimg + packed_word(f / kPackedFrames, nblk, k / kPackedComps,
                                           k % kPackedComps, f6 & ~1);
    float v[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q)
        v[q] = (f6 & 1) ? bf16_hi(w[q * kPackedPlaneWords]) : bf16_lo(w[q * kPackedPlaneWords]);
    R[idx] = (v[1] + v[2]) + v[0];
}

// float32 responsibilities [T, K] (times state responsibilities [T, S]) -> packed
// tiles: one workgroup per tile of 64 frames x 128 components, transposed through
// LDS (rows of 129 words: conflict-free both ways), split, stored as whole
// 16-byte chunks.
__global__ __launch_bounds__(256) void pack_resps_kernel(int64_t nframes, int K, int S, int G,
                                                         const float* __restrict__ R,
                                                         const float* __restrict__ SR,
                                                         unsigned* __restrict__ out) {
    __shared__ float tile[kPackedFrames * 129];
    const int64_t tau = blockIdx.x, t0 = tau * kPackedFrames;
    const int beta = blockIdx.y, nblk = gridDim.y, kc0 = beta * kPackedComps;
    const int rows = (int)(nframes - t0 < kPackedFrames ? nframes - t0 : kPackedFrames);
    for (int e = threadIdx.x; e < kPackedFrames * (kPackedComps / 4); e += 256) {
        const int f = e / (kPackedComps / 4), c4 = 4 * (e - f * (kPackedComps / 4)), k = kc0 + c4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (f < rows && k < K) {                               // K % 4 == 0
            v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(R + (t0 + f) * K + k));
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
    for (int e = threadIdx.x; e < kPackedComps * 8; e += 256) {
        const int kk = e >> 3, c8 = e & 7;
        u4 o[NP];
#pragma unroll
        for (int wd = 0; wd < 4; ++wd) {
            unsigned w3[3];
            split3(tile[(8 * c8 + 2 * wd) * 129 + kk], tile[(8 * c8 + 2 * wd + 1) * 129 + kk], w3);
#pragma unroll
            for (int q = 0; q < NP; ++q) o[q][wd] = w3[q];
        }
        unsigned* dst = out + packed_word(tau, nblk, beta, kk, 8 * c8);
#pragma unroll
        for (int q = 0; q < NP; ++q)
            __builtin_nontemporal_store(o[q], reinterpret_cast<u4*>(dst + q * kPackedPlaneWords));
    }
}

// ---------------------------------------------------------------------------
// Fused accumulation for mixture sets with few statistics per Gaussian (diagonal
// / isotropic covariances): S[k, q] += sum_t r[t,k] sr[t, k / G] PHI_q(x_t) WITHOUT
// the responsibilities in memory.  The E-step only leaves the per-state
// log-normalisers [T, S]; after the forward-backward pass this kernel recomputes
// the component logits of a tile of 32 frames x 16 NTC components on the matrix
// cores (the k-loop of llhx_kernel: same packed parameters, same products in the
// same order, so that exp(l - log_norm) is exactly pass 1's responsibility), turns
// them into r sr = exp(l - log_norm[t, s]) sr[t, s] in registers -- no maximum, no
// sum: the normaliser is known -- and feeds them straight back to the matrix cores
// as the A operand of the statistics product: the C layout of the logits (lane
// (i, g): component i, frames 4g..4g+3 of both 16-frame tiles) IS the A layout of a
// 16 x 32 [component x frame] operand when the 32 frames of the contraction are
// taken in the order (tile 0: 4g..4g+3, tile 1: 4g..4g+3), and the B operand
// PHI_q(x_f) is generated from the transposed frame tile in that same order.
// A wave keeps its 16 NTC x 16 NQT statistics tile in registers over all the
// frame tiles it walks (fp32, <= 4096 frames), then adds it to the fp64 image.
// ---------------------------------------------------------------------------
constexpr int kAfXS = 36;                 // row stride (floats) of the transposed frame tile
#ifndef BEER_AF_MAXFRAMES
#define BEER_AF_MAXFRAMES 2048
#endif
constexpr int kAfMaxFramesPerWave = BEER_AF_MAXFRAMES; // MFMA accumulations per sum: 64 (half of
// the packed accumulation's 128, BEER_OPT_AX_MAXFRAMES; 1024 -> 2048 frames: fewer flushes, 29.8 -> 29.35 ms
// per 10 M frames at config 3, count conservation unchanged at 3.6e-8)

// BLK: the states of a 64-component chunk are at most 4 consecutive ones (G >= 16): a
// lane fetches the 4 normalisers or posteriors of one frame row in one go, ONE TILE
// AHEAD (as it does the frames), parks them in LDS when their tile starts and the
// exponentials read them from there -- instead of 16 four-byte gathers per lane whose
// latency every tile waited for.
template <int NTC, int NQT, bool G4, int WAVES, int kXP, bool BLK>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void accf_kernel(
    int64_t nframes, int D, int K, int S, int G, int Greal, int nk, int nslab,
    const float* __restrict__ X, const u4* __restrict__ Pall, const int* __restrict__ tab,
    const float* __restrict__ log_norm, const float* __restrict__ sr,
    int64_t frames_per_block, double* __restrict__ Sp, const float* __restrict__ c0p) {
    constexpr int MT = 2, FW = 32, QT = NTC / 4, NTHREADS = 64 * WAVES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int64_t bx;
    int by;
    {
        const int nch = (K + 16 * NTC - 1) / (16 * NTC);
        if (!xcd_block((nframes + frames_per_block - 1) / frames_per_block, nch, nch, bx, by))
            return;
    }
    const int D4 = d4_of(D), Dp = 4 * D4, LD = ld16_of(D), nq = nslab * 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int xt_floats = (D + 2) * kAfXS;
    const int nk_used = (nslab + 7) / 8;
    // LDS: the chunk's packed parameters (nk_used x NTC blocks of 3 KiB, read by every
    // wave for every frame tile), the slab table, then per wave the frame tile
    // row-major (A fragments of the logits) and transposed (B fragments of the
    // statistics)
    const int p_u4 = nk_used * NTC * kBlockU4;
    u4* Ps = reinterpret_cast<u4*>(smem);
    int* tabs = reinterpret_cast<int*>(Ps + p_u4);
    constexpr int kLs = BLK ? 2 * FW * 4 : 0;          // [ln | sr][row][4 states]
    int* offs = tabs + (nk + 1) * 8;                   // [2 NQT][64 lanes]: see xa_off below
    float* xw = reinterpret_cast<float*>(offs + 2 * NQT * 64) + wave * (FW * LD + xt_floats + kLs);
    float* xt = xw + FW * LD;
    float* lsw = xt + xt_floats;
    {
        const u4* src = Pall + (size_t)by * nk * NTC * kBlockU4;
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
    const float c0 = c0p[0];
    const int64_t tb = bx * frames_per_block;
    const int64_t te = tb + frames_per_block < nframes ? tb + frames_per_block : nframes;
    const u4* Pl = Ps + lane;
    const int* tl = tabs + 2 * g;
    const float* xrow[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) xrow[m] = xw + (m * 16 + i) * LD;

    auto factors = [&](int uu, int& a, int& b) {
        const int col = 16 * uu + i, slab = col >> 2;
        a = b = Dp + 2;
        if (slab < nslab) {
            const int t = tabs[slab];
            b = ((t >> 8) & 0xff) + (col & 3);
            a = (t >> 16) ? b : (t & 0xff);
        }
    };
    // offsets (floats, in the transposed tile) of the two factors of the lane's column of
    // statistic tile uu: a lane-indexed table in LDS, read back where it is used (12
    // registers that the next tile's frames now wait in)
    if (wave == 0) {
#pragma unroll
        for (int uu = 0; uu < NQT; ++uu) {
            int a, b;
            factors(uu, a, b);
            offs[(2 * uu) * 64 + lane] = (a < D ? a : (a == Dp ? D : D + 1)) * kAfXS + 4 * g;
            offs[(2 * uu + 1) * 64 + lane] = (b < D ? b : (b == Dp ? D : D + 1)) * kAfXS + 4 * g;
        }
    }
    __syncthreads();
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

    // BLK: the 4 states fetched per row start at st0c; the lane's state is st0c + sidx
    int st0c = 0, sidx = 0;
    if (BLK) {
        const int st0 = kbase / G < S ? kbase / G : S - 1;
        st0c = st0 < S - 4 ? st0 : S - 4;
        sidx = st_of[0][0] - st0c;
    }

    f32x4 sacc[NTC][NQT];
#pragma unroll
    for (int c = 0; c < NTC; ++c)
#pragma unroll
        for (int uu = 0; uu < NQT; ++uu) sacc[c][uu] = f32x4{0, 0, 0, 0};

    struct AFrag { u4 w[NP][MT]; };
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
            unsigned w3[3];
            split3(p[2 * e], p[2 * e + 1], w3);
#pragma unroll
            for (int q = 0; q < NP; ++q) f.w[q][m][2 * h + e] = w3[q];
        }
    };

    const int C4 = D >> 2;
    const bool rows4 = (D & 3) == 0;
    const f32x4* X4 = reinterpret_cast<const f32x4*>(X);
    // lane = (row r, half h of the row), pieces h, h + 2, h + 4 .. of it: every address
    // below is ONE per-lane base plus a compile-time constant per piece.  Rows past the
    // end are read from the last row: their responsibilities get weight 0 below.
    const int NPC = (C4 + 1) >> 1, lr = lane & 31, lh = lane >> 5;
    float* xw_l = xw + lr * LD + 4 * lh;                       // + 8 it
    float* xt_l = xt + 4 * lh * kAfXS + lr;                    // + (8 it + j) kAfXS
    // The loads of a tile (frames; BLK: normalisers / posteriors) into registers, rows
    // clamped to the block's last frame
    f32x4 xv[kXP], lsv = f32x4{0, 0, 0, 0};
    auto issue = [&](int64_t fbn) {
        const int rows_n = (int)(te - fbn < FW ? te - fbn : FW);
        const int64_t rown = fbn + (lr < rows_n ? lr : rows_n - 1);
        if (rows4) {
            const f32x4* src = X4 + rown * C4;
#pragma unroll
            for (int it = 0; it < kXP; ++it) {
                const int pc = lh + 2 * it;
                xv[it] = src[pc < C4 ? pc : C4 - 1];
            }
        }
        if (BLK) {
            const float* src = ((lh && sr) ? sr : log_norm) + rown * S + st0c;
#pragma unroll
            for (int j = 0; j < 4; ++j) lsv[j] = src[j];
        }
    };
    const int64_t fb0 = tb + (int64_t)wave * FW;
    if (fb0 < te) issue(fb0);
    // frame tiles of this wave: tb + 32 (wave + WAVES n)
    for (int64_t fb = fb0; fb < te; fb += WAVES * FW) {
        const int rows = (int)(te - fb < FW ? te - fb : FW);              // >= 1
        // ---- the frame tile, row-major and transposed (wave-private LDS) ----
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        if (rows4) {
#pragma unroll
            for (int it = 0; it < kXP; ++it) {
                if (it >= NPC) break;
                if (lh + 2 * it < C4) {
                    const f32x4 v = xv[it];
                    *reinterpret_cast<f32x4*>(xw_l + 8 * it) = v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) xt_l[(8 * it + j) * kAfXS] = v[j];
                }
            }
        } else {
            const float* Xt = X + fb * D;
            for (int idx = lane; idx < FW * D; idx += 64) {
                const int r = idx / D, c = idx - r * D;
                const float v = r < rows ? Xt[idx] : 0.f;
                xw[r * LD + c] = v;
                xt[c * kAfXS + r] = v;
            }
        }
        // the per-state normalisers and posteriors of the tile's rows: (frame 16 m + 4 g
        // + r, state of the lane's components)
        float nl2[QT][MT][4][NST], wg[QT][MT][4][NST];
        bool skip = false;
        if (BLK) {
            *reinterpret_cast<f32x4*>(lsw + (lh * FW + lr) * 4) = lsv;
            // A tile none of whose frames gives the chunk's states any posterior
            // contributes exactly nothing: skip it (alignment graphs: most of the model's
            // states are absent from an utterance, their posteriors are exact zeros)
            if (sr && !(BEER_AF_ABL & 64)) {
                float any = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) any = __builtin_fmaxf(any, __builtin_fabsf(lsv[j]));
                skip = __builtin_amdgcn_ballot_w64(lh == 1 && lr < rows && any != 0.f) == 0;
            }
            // the next tile's loads fly while this one is worked on
            if (fb + WAVES * FW < te) issue(fb + WAVES * FW);
            if (skip) continue;
        } else {
            // (unconditional loads on clamped rows, validity applied to constants: a select
            // or a branch on the loaded value makes hipcc load and wait one at a time)
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
                        for (int j = 0; j < NST; ++j) {
                            nl2[q][m][r][j] = ln_t[rc * S + st_of[q][j]];
                            wg[q][m][r][j] = sr ? sr_t[rc * S + st_of[q][j]] : 1.f;
                        }
                    }
            if (fb + WAVES * FW < te) issue(fb + WAVES * FW);
            if (sr && !(BEER_AF_ABL & 64)) {
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
                if (__builtin_amdgcn_ballot_w64(any != 0.f) == 0) continue;
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
        // (A fragments of a k-step are built right before its MFMAs: with two waves per
        // SIMD the other wave's MFMAs cover the arithmetic, and a second set of fragments
        // in flight cost 24 registers -- the ones the next tile's frames now wait in)
        for (int s = 0; s < nk_used; ++s) {
            AFrag cur;
#pragma unroll
            for (int hh = 0; hh < MT * 2; ++hh) make_half(s, hh % MT, hh / MT, cur);
#pragma unroll
            for (int c = 0; c < NTC; ++c) {
                u4 bp[NP];
#pragma unroll
                for (int pq = 0; pq < NP; ++pq) bp[pq] = Pl[(s * NTC + c) * kBlockU4 + 64 * pq];
                // (the products in the order of llhx_kernel: same roundings)
#pragma unroll
                for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        acc[m][c] = mfma_bf16(cur.w[kProdA[pr]][m], bp[kProdB[pr]], acc[m][c]);
            }
        }

        if (BLK) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * m + 4 * g + r;
                    nl2[0][m][r][0] = lsw[row * 4 + sidx];
                    wg[0][m][r][0] = sr ? lsw[(FW + row) * 4 + sidx] : 1.f;
                }
        }
        // rows past the end: weight 0, and a normaliser that keeps the exponential at 0
        // (0 x inf would be NaN)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // (the recomputed logits lack the common constant c0, see const_max_kernel)
                const bool ok = 16 * m + 4 * g + r < rows;
                const float pen = ok ? c0 : -1.0e30f, mult = ok ? 1.f : 0.f;
#pragma unroll
                for (int q = 0; q < QT; ++q)
#pragma unroll
                    for (int j = 0; j < NST; ++j) {
                        nl2[q][m][r][j] = pen - nl2[q][m][r][j];
                        wg[q][m][r][j] *= mult;
                    }
            }
        // ---- r sr = exp(l - log_norm) sr, split into the A fragments ----
        // A fragment of component tile nt: words 0, 1 = frames 4g..4g+3 of tile 0,
        // words 2, 3 = the same rows of tile 1
        u4 ar[NTC][NP];
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
                    v[r] = (BEER_AF_ABL & 8) ? acc[m][nt][r] + nl2[q][m][r][jj] + wg[q][m][r][jj] :
                           __builtin_amdgcn_exp2f((acc[m][nt][r] + nl2[q][m][r][jj]) *
                                                  1.44269504088896340736f) *
                           wg[q][m][r][jj];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    unsigned w3[3];
                    if (BEER_AF_ABL & 8) {
                        w3[0] = __builtin_bit_cast(unsigned, v[2 * e]);
                        w3[1] = __builtin_bit_cast(unsigned, v[2 * e + 1]);
                        w3[2] = w3[0] ^ w3[1];
                    } else
                    split3(v[2 * e], v[2 * e + 1], w3);
#pragma unroll
                    for (int pq = 0; pq < NP; ++pq) ar[nt][pq][2 * m + e] = w3[pq];
                }
            }
        }

        // ---- statistics: sacc[c][uu] += A'(c) x B'(uu) ----
        auto gen_b = [&](int uu, u4 (&out)[NP]) {
            const int xa_off = offs[(2 * uu) * 64 + lane], xb_off = offs[(2 * uu + 1) * 64 + lane];
            const f32x4 xa0 = *reinterpret_cast<const f32x4*>(xt + xa_off);
            const f32x4 xa1 = *reinterpret_cast<const f32x4*>(xt + xa_off + 16);
            const f32x4 xb0 = *reinterpret_cast<const f32x4*>(xt + xb_off);
            const f32x4 xb1 = *reinterpret_cast<const f32x4*>(xt + xb_off + 16);
            const f32x4 p0 = xa0 * xb0, p1 = xa1 * xb1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned w3[3];
                split3(e < 2 ? p0[2 * e] : p1[2 * e - 4], e < 2 ? p0[2 * e + 1] : p1[2 * e - 3],
                       w3);
#pragma unroll
                for (int pq = 0; pq < NP; ++pq) out[pq][e] = w3[pq];
            }
        };
        u4 bq[2][NP];
        gen_b(0, bq[0]);
#pragma unroll
        for (int uu = 0; uu < NQT; ++uu) {
            const int cur = uu & 1;
            if (uu + 1 < NQT) {
                if (BEER_AF_ABL & 16) {
#pragma unroll
                    for (int pq = 0; pq < NP; ++pq) bq[cur ^ 1][pq] = bq[cur][pq];
                } else
                gen_b(uu + 1, bq[cur ^ 1]);
            }
#pragma unroll
            for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                for (int c = 0; c < NTC; ++c)
                    sacc[c][uu] = mfma_bf16(ar[c][kProdA[pr]], bq[cur][kProdB[pr]], sacc[c][uu]);
        }
    }

    // ---- flush: rows = components kbase + 64 (c / 4) + 4 (4 g + r) + c % 4 ----
    if (BEER_AF_ABL & 1) {
        float t = 0.f;
#pragma unroll
        for (int uu = 0; uu < NQT; ++uu)
#pragma unroll
            for (int c = 0; c < NTC; ++c) t += sacc[c][uu][0] + sacc[c][uu][1] + sacc[c][uu][2] + sacc[c][uu][3];
        if (t == 1.2345f) Sp[0] = 1.0;
        return;
    }
    // The WAVES waves of the workgroup hold sums over different frames of the same 64
    // components: added up through LDS (the parameters' region, no longer needed; fp64)
    // one statistic tile at a time, so that the fp64 image sees one atomic per
    // workgroup and element instead of one per wave.
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);          // [WAVES][16 = c * 4 + r][64 lanes]
    constexpr int EPT = 16 * 64 / NTHREADS;               // elements per thread and tile
    static_assert(16 * 64 % NTHREADS == 0, "whole elements per thread");
    int64_t dst_row[EPT];
    int dst_i[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int el = tid + e * NTHREADS, j = el >> 6, ln = el & 63;
        const int c = j >> 2, r = j & 3, gg = ln >> 4;
        const int slot = kbase + 64 * (c >> 2) + 4 * (4 * gg + r) + (c & 3);
        // slot -> component (padded slots of a group and slots past the end: none)
        const int gi = slot % G;
        dst_row[e] = slot < K && gi < Greal ? (int64_t)((slot / G) * Greal + gi) * nq : -1;
        dst_i[e] = ln & 15;
    }
#pragma unroll
    for (int uu = 0; uu < NQT; ++uu) {
#pragma unroll
        for (int c = 0; c < NTC; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(wave * 16 + c * 4 + r) * 64 + lane] = sacc[c][uu][r];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int el = tid + e * NTHREADS;
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) t += (double)red[w * 1024 + el];
            const int q = 16 * uu + dst_i[e];
            if (dst_row[e] >= 0 && q < nq) atomicAdd(Sp + dst_row[e] + q, t);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// Frame fragment images.  The fused accumulation is bound by VECTOR INSTRUCTION ISSUE,
// not by the matrix pipe (PMC round 3: 1050 VALU + 288 MFMA instructions per wave-tile at
// 4 issue cycles each is more than the 4608 cycles the MFMAs execute for; no schedule
// -- two waves per SIMD, one wave hand-interleaved -- can go below the issue count).  700
// of the 1050 VALU instructions rebuild operands that depend on the frames only: the A
// fragments of the logits and the B fragments of the statistics, identical for all 30
// component chunks of K = 1920 and for every VB iteration over the same frames.
// frame_image_kernel builds them once per frame tile, in exactly the registers' layout:
//   img[tile of 32 frames][ k-step s ][ piece q ][ frame tile m ][ 64 lanes ] u4   (A)
//   img[tile            ][ NKU*6 +  statistic tile uu * 3 + piece q ][ 64 lanes ] u4   (B)
// (1056 B per frame at D = 40: 3.5 GB per 3.33 M frames, read by the 30 chunk blocks of
// a frame block that xcd_block() puts on one XCD next to each other.)  accfi_kernel is
// accf_kernel with every fragment a 16-byte load.
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(256) void frame_image_kernel(int64_t nframes, int D, int nk, int nslab,
                                                          int nqt, const float* __restrict__ X,
                                                          const int* __restrict__ tab,
                                                          u4* __restrict__ img) {
    constexpr int MT = 2, FW = 32, NW = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int D4 = d4_of(D), Dp = 4 * D4, LD = ld16_of(D);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int nk_used = (nslab + 7) / 8, xt_floats = (D + 2) * kAfXS;
    int* tabs = reinterpret_cast<int*>(smem);
    float* xw = reinterpret_cast<float*>(tabs + (nk + 1) * 8) + wave * (FW * LD + xt_floats);
    float* xt = xw + FW * LD;
    for (int idx = tid; idx < (nk + 1) * 8; idx += 256) tabs[idx] = tab[idx];
    const int64_t tile = (int64_t)blockIdx.x * NW + wave, fb = tile * FW;
    stage_rows<FW>(X, fb < nframes ? fb : 0, nframes, D, LD, lane, xw);   // (a wave past the end stages tile 0)
    __syncthreads();
    if (fb >= nframes) return;
    for (int idx = lane; idx < (D + 2) * FW; idx += 64) {
        const int c = idx / FW, r = idx - c * FW;
        xt[c * kAfXS + r] = c < D ? xw[r * LD + c] : (c == D ? 1.f : 0.f);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    u4* out = img + tile * frame_image_tile_u4(nk_used, nqt) + lane;
    const int* tl = tabs + 2 * g;
    for (int s = 0; s < nk_used; ++s) {
        u4 w[NP][MT];
#pragma unroll
        for (int hh = 0; hh < 2 * MT; ++hh) {
            const int m = hh % MT, h = hh / MT;
            const float* xrow = xw + (m * 16 + i) * LD;
            const int t = tl[8 * s + h];
            const int a = t & 0xff, j = (t >> 8) & 0xff;
            const bool sq = (t >> 16) != 0;
            const f32x4 bb = *reinterpret_cast<const f32x4*>(xrow + j);
            const float xx = xrow[a];
            f32x4 p;
#pragma unroll
            for (int e = 0; e < 4; ++e) p[e] = bb[e] * (sq ? bb[e] : xx);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                unsigned w3[3];
                split3(p[2 * e], p[2 * e + 1], w3);
#pragma unroll
                for (int q = 0; q < NP; ++q) w[q][m][2 * h + e] = w3[q];
            }
        }
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int m = 0; m < MT; ++m) out[((s * NP + q) * MT + m) * 64] = w[q][m];
    }
    out += (size_t)nk_used * NP * MT * 64;
    const int ndata = 2 * D4;                              // data slabs: squares and linear terms
    for (int uu = 0; uu < nqt; ++uu) {
        const int col = 16 * uu + i, slab = (col >> 2) < ndata ? stat_slab(col >> 2) : nslab;
        int a = Dp + 2, b = Dp + 2;
        if (slab < nslab) {
            const int t = tabs[slab];
            b = ((t >> 8) & 0xff) + (col & 3);
            a = (t >> 16) ? b : (t & 0xff);
        }
        const int xa_off = (a < D ? a : (a == Dp ? D : D + 1)) * kAfXS + 4 * g;
        const int xb_off = (b < D ? b : (b == Dp ? D : D + 1)) * kAfXS + 4 * g;
        const f32x4 xa0 = *reinterpret_cast<const f32x4*>(xt + xa_off);
        const f32x4 xa1 = *reinterpret_cast<const f32x4*>(xt + xa_off + 16);
        const f32x4 xb0 = *reinterpret_cast<const f32x4*>(xt + xb_off);
        const f32x4 xb1 = *reinterpret_cast<const f32x4*>(xt + xb_off + 16);
        const f32x4 p0 = xa0 * xb0, p1 = xa1 * xb1;
        u4 o[NP];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            unsigned w3[3];
            split3(e < 2 ? p0[2 * e] : p1[2 * e - 4], e < 2 ? p0[2 * e + 1] : p1[2 * e - 3], w3);
#pragma unroll
            for (int q = 0; q < NP; ++q) o[q][e] = w3[q];
        }
#pragma unroll
        for (int q = 0; q < NP; ++q) out[(uu * NP + q) * 64] = o[q];
    }
}

// NS: the states a 64-component chunk may reach into, a block of NS consecutive ones whose
// normalisers / posteriors a wave stages per tile -- 4 (groups of >= 16 Gaussians: config 3) or 16
// (groups of 4 .. 12: the recipes' 4 and 10 Gaussians per state).
template <int NKU, int WAVES = 8, int NS = 4>
__global__ __launch_bounds__(64 * WAVES, 2) void accfi_kernel(
    int64_t nframes, int K, int S, int G, int Greal, int nk, int nslab,
    const u4* __restrict__ img, const u4* __restrict__ Pall, const float* __restrict__ log_norm,
    const float* __restrict__ sr, int64_t frames_per_block, double* __restrict__ Sp,
    const float* __restrict__ c0p) {
    constexpr int NTC = 4, NQT = img_nqt(NKU), MT = 2, FW = 32, NTHREADS = 64 * WAVES;
    // (the 6-tile format leaves no registers for a third B fragment in flight)
    constexpr int NBQ = NQT > 5 ? 2 : 3;
    constexpr int kTileU4 = (NKU * NP * MT + NQT * NP) * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int64_t bx;
    int by;
    {
        const int nch = (K + 16 * NTC - 1) / (16 * NTC);
        if (!xcd_block((nframes + frames_per_block - 1) / frames_per_block, nch, nch, bx, by))
            return;
    }
    const int nq = nslab * 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    // LDS: the chunk's packed parameters, then per wave the normalisers / posteriors of a tile
    constexpr int p_u4 = NKU * NTC * kBlockU4;
    u4* Ps = reinterpret_cast<u4*>(smem);
    float* lsw = reinterpret_cast<float*>(Ps + p_u4) + wave * (2 * FW * NS);
    {
        const u4* src = Pall + (size_t)by * nk * NTC * kBlockU4;
        for (int idx = tid; idx < p_u4; idx += NTHREADS) Ps[idx] = src[idx];
    }
    __syncthreads();
    if (BEER_ACCFI_SLEEP > 0 && (WAVES == 8 ? wave >= 4 : (blockIdx.x >> 3) & 1)) {
#pragma unroll
        for (int n = 0; n < (BEER_ACCFI_SLEEP + 126) / 127; ++n)
            __builtin_amdgcn_s_sleep(BEER_ACCFI_SLEEP < 127 ? BEER_ACCFI_SLEEP : 127);
    }
    const int kbase = by * (16 * NTC);
    const float c0 = c0p[0];
    const int64_t tb = bx * frames_per_block;
    const int64_t te = tb + frames_per_block < nframes ? tb + frames_per_block : nframes;
    const u4* Pl = Ps + lane;
    int sidx, st0c, smax;
    {
        const int s0 = (kbase + 4 * i) / G, st = s0 < S ? s0 : S - 1;
        const int st0 = kbase / G < S ? kbase / G : S - 1;
        st0c = st0 < S - NS ? st0 : (S > NS ? S - NS : 0);
        sidx = st - st0c;                                  // (< NS: the host checked the chunks)
        smax = (S - st0c < NS ? S - st0c : NS) - 1;        // fewer than NS states in all: the last again
    }
    f32x4 sacc[NTC][NQT];
#pragma unroll
    for (int c = 0; c < NTC; ++c)
#pragma unroll
        for (int uu = 0; uu < NQT; ++uu) sacc[c][uu] = f32x4{0, 0, 0, 0};
    // counts N_k = sum_t r sr of the lane's component (tile c, column i) over the frames of
    // its row group: plain float32 additions of the weights the statistics are multiplied with
    float cnt[NTC];
#pragma unroll
    for (int c = 0; c < NTC; ++c) cnt[c] = 0.f;

    const int lr = lane & 31, lh = lane >> 5;
    float lsv[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) lsv[j] = 0.f;
    auto issue = [&](int64_t fbn) {
        const int rows_n = (int)(te - fbn < FW ? te - fbn : FW);
        const int64_t rown = fbn + (lr < rows_n ? lr : rows_n - 1);
        const float* src = ((lh && sr) ? sr : log_norm) + rown * S + st0c;
#pragma unroll
        for (int j = 0; j < NS; ++j) lsv[j] = src[j < smax ? j : smax];
    };
    const int64_t fb0 = tb + (int64_t)wave * FW;
    if (fb0 < te) issue(fb0);
    // The fragment loads run far ahead of their MFMAs (L2 latency, 40 % of the wave time
    // when they were issued one batch ahead): a tile's first A fragments during the
    // statistics of the tile before, its first two B fragments when its logits start.
    u4 af[2][NP][MT];
    u4 bq[NBQ][NP];
    bool a0_ready = false;
    for (int64_t fb = fb0; fb < te; fb += WAVES * FW) {
        const int rows = (int)(te - fb < FW ? te - fb : FW);
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < NS; j += 4)
            *reinterpret_cast<f32x4*>(lsw + (lh * FW + lr) * NS + j) =
                f32x4{lsv[j], lsv[j + 1], lsv[j + 2], lsv[j + 3]};
        bool skip = false;
        if (sr) {
            float any = 0.f;
#pragma unroll
            for (int j = 0; j < NS; ++j) any = __builtin_fmaxf(any, __builtin_fabsf(lsv[j]));
            skip = __builtin_amdgcn_ballot_w64(lh == 1 && lr < rows && any != 0.f) == 0;
        }
        if (fb + WAVES * FW < te) issue(fb + WAVES * FW);
        if (skip) {
            a0_ready = false;
            continue;
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        const u4* ti = img + (fb / FW) * (int64_t)kTileU4 + lane;
        const bool has_next = fb + WAVES * FW < te;
        const u4* tn = ti + (has_next ? (int64_t)WAVES * kTileU4 : 0);

        // ---- logits of 32 frames x 64 components: A fragments straight from the image ----
        f32x4 acc[MT][NTC];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int c = 0; c < NTC; ++c) acc[m][c] = f32x4{0, 0, 0, 0};
        if (!a0_ready) {
#pragma unroll
            for (int q = 0; q < NP; ++q)
#pragma unroll
                for (int m = 0; m < MT; ++m) af[0][q][m] = ti[(q * MT + m) * 64];
        }
        // the first B fragment of the statistics (the second when the logits are done: the
        // exponentials cover its latency, and the logits phase has no registers to spare)
#pragma unroll
        for (int q = 0; q < NP; ++q) bq[0][q] = ti[(NKU * NP * MT + q) * 64];
        // (B fragments of the logits: the three planes of component tile c + 1 are read from
        // LDS at the START of tile c's 12 MFMAs -- see lnfi_kernel)
        u4 bp[2][NP];
#pragma unroll
        for (int pq = 0; pq < NP; ++pq) bp[0][pq] = Pl[64 * pq];
        __builtin_amdgcn_sched_group_barrier(0x100, NP, 0);
#pragma unroll
        for (int s = 0; s < NKU; ++s) {
            if (s + 1 < NKU) {
#pragma unroll
                for (int q = 0; q < NP; ++q)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        af[(s + 1) & 1][q][m] = (BEER_AFI_ABL & 4) ? af[s & 1][q][m] :
                                                ti[(((s + 1) * NP + q) * MT + m) * 64];
            }
            // the loads of this k-step first (the counts must match what is issued here, or
            // hipcc moves other loads in to fill the group)
            if (s == 0) __builtin_amdgcn_sched_group_barrier(0x020, NKU > 1 ? 9 : 3, 0);
            else if (s + 1 < NKU) __builtin_amdgcn_sched_group_barrier(0x020, 6, 0);
#pragma unroll
            for (int c = 0; c < NTC; ++c) {
                constexpr int kLast = NKU * NTC - 1;
                const int gi = s * NTC + c, gn = gi < kLast ? gi + 1 : kLast;
#pragma unroll
                for (int pq = 0; pq < NP; ++pq) bp[(gi + 1) & 1][pq] = Pl[gn * kBlockU4 + 64 * pq];
                // (the products in the order of llhx_kernel: same roundings)
#pragma unroll
                for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        acc[m][c] = mfma_bf16(af[s & 1][kProdA[pr]][m], bp[gi & 1][kProdB[pr]], acc[m][c]);
                __builtin_amdgcn_sched_group_barrier(0x100, NP, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 6 * MT, 0);
            }
        }

        // ---- r sr = exp(l - log_norm) sr, split into the A fragments of the statistics ----
        if constexpr (NBQ == 3) {
#pragma unroll
            for (int q = 0; q < NP; ++q) bq[1][q] = ti[(NKU * NP * MT + NP + q) * 64];
        }
        u4 ar[NTC][NP];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * m + 4 * g + r;
                const bool ok = row < rows;
                // (the recomputed logits lack the common constant c0; rows past the end:
                // weight 0 and a normaliser that keeps the exponential at 0)
                const float nl2 = (ok ? c0 : -1.0e30f) - lsw[row * NS + sidx];
                const float wg = (sr ? lsw[(FW + row) * NS + sidx] : 1.f) * (ok ? 1.f : 0.f);
#pragma unroll
                for (int nt = 0; nt < NTC; ++nt)
                    acc[m][nt][r] = exp2_valu((acc[m][nt][r] + nl2) * 1.44269504088896340736f) * wg;
            }
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt)
            cnt[nt] += ((acc[0][nt][0] + acc[0][nt][1]) + (acc[0][nt][2] + acc[0][nt][3])) +
                       ((acc[1][nt][0] + acc[1][nt][1]) + (acc[1][nt][2] + acc[1][nt][3]));
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    unsigned w3[3];
                    split3(acc[m][nt][2 * e], acc[m][nt][2 * e + 1], w3);
#pragma unroll
                    for (int pq = 0; pq < NP; ++pq) ar[nt][pq][2 * m + e] = w3[pq];
                }

        // ---- statistics: sacc[c][uu] += A'(c) x B'(uu), B' from the image two tiles ahead ----
#pragma unroll
        for (int uu = 0; uu < NQT; ++uu) {
            if (uu + NBQ - 1 < NQT) {
#pragma unroll
                for (int q = 0; q < NP; ++q)
                    bq[(uu + NBQ - 1) % NBQ][q] = (BEER_AFI_ABL & 1) ? bq[uu % NBQ][q] :
                                                  ti[(NKU * NP * MT + (uu + NBQ - 1) * NP + q) * 64];
            }
            if (uu == NQT - 3) {
                // (no next tile: this one again, harmlessly)
#pragma unroll
                for (int q = 0; q < NP; ++q)
#pragma unroll
                    for (int m = 0; m < MT; ++m) af[0][q][m] = tn[(q * MT + m) * 64];
                a0_ready = has_next;
            }
#pragma unroll
            for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                for (int c = 0; c < NTC; ++c)
                    sacc[c][uu] = mfma_bf16(ar[c][kProdA[pr]], bq[uu % NBQ][kProdB[pr]], sacc[c][uu]);
            if (uu == NQT - 3) __builtin_amdgcn_sched_group_barrier(0x020, 9, 0);
            else if (uu + NBQ - 1 < NQT) __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 6 * NTC, 0);
        }
    }

    // ---- flush: the waves' partial sums through LDS (fp64), one atomic per element ----
    if (BEER_AFI_ABL & 2) {
        float t = 0.f;
#pragma unroll
        for (int uu = 0; uu < NQT; ++uu)
#pragma unroll
            for (int c = 0; c < NTC; ++c) t += sacc[c][uu][0] + sacc[c][uu][1] + sacc[c][uu][2] + sacc[c][uu][3];
        if (t == 1.2345f) Sp[0] = 1.0;
        return;
    }
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);          // [WAVES][16 = c * 4 + r][64 lanes]
    const int ndata = nslab - 1 - (nslab - 1) / 8;         // squares and linear terms (no constants)
    constexpr int EPT = 16 * 64 / NTHREADS;
    int64_t dst_row[EPT];
    int dst_i[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int el = tid + e * NTHREADS, j = el >> 6, ln = el & 63;
        const int c = j >> 2, r = j & 3, gg = ln >> 4;
        const int slot = kbase + 64 * (c >> 2) + 4 * (4 * gg + r) + (c & 3);
        const int gi = slot % G;
        dst_row[e] = slot < K && gi < Greal ? (int64_t)((slot / G) * Greal + gi) * nq : -1;
        dst_i[e] = ln & 15;
    }
#pragma unroll
    for (int uu = 0; uu < NQT; ++uu) {
#pragma unroll
        for (int c = 0; c < NTC; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(wave * 16 + c * 4 + r) * 64 + lane] = sacc[c][uu][r];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int el = tid + e * NTHREADS;
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) t += (double)red[w * 1024 + el];
            // statistic column of (tile uu, column i): data slab (16 uu + i) / 4 of the table
            const int dslab = (16 * uu + dst_i[e]) >> 2;
            const int q = stat_slab(dslab) * 4 + (dst_i[e] & 3);
            if (dst_row[e] >= 0 && dslab < ndata) atomicAdd(Sp + dst_row[e] + q, t);
        }
        __syncthreads();
    }
    // the counts: over the lane's 4 row groups (lanes i, i + 16, i + 32, i + 48), then over the
    // waves in fp64, into entry 0 of the final constant slab (what unpack_kernel reads)
#pragma unroll
    for (int c = 0; c < NTC; ++c) {
        float v = cnt[c];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (lane < 16) red[(wave * NTC + c) * 16 + lane] = v;
    }
    __syncthreads();
    if (tid < 16 * NTC) {
        const int c = tid >> 4, ii = tid & 15;
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) t += (double)red[(w * NTC + c) * 16 + ii];
        const int slot = kbase + 4 * ii + c, gi = slot % G;
        if (slot < K && gi < Greal)
            atomicAdd(Sp + (int64_t)((slot / G) * Greal + gi) * nq + (nslab - 1) * 4, t);
    }
}

// ---------------------------------------------------------------------------
// Log-normalisers of a mixture set from a frame fragment image (pass 1 of the config-3
// iteration; replaces llhx_kernel<.., LNO, IMG> where the packed image is lane-major).
// That kernel gave every wave ONE 32-frame tile and streamed the chunk's packed
// parameters (3 k-steps x 16 tiles x 3 KiB = 144 KiB at D = 40) from L2 for each of
// them, one batch of two tiles ahead: 80 s_waitcnt per 384 MFMAs in its loop, the
// vector-memory path of a CU more than half busy with re-reading the same 144 KiB, 58 %
// MFMA busy.  Here a workgroup of 8 waves (two per SIMD) keeps the WHOLE chunk in LDS --
// 144 KiB of the CU's 160 -- and walks a block of frames: B fragments are ds_read_b128
// with LDS latency, the only global loads left are the 18 A fragments per tile (from the
// image, a k-step ahead; the next tile's first k-step during the epilogue).  Same
// products in the same order as llhx_kernel and accfi_kernel: bit-identical logits.
// ---------------------------------------------------------------------------
// NT: component tiles per chunk -- 16 (256 components: 48 KiB of LDS per k-step) up to three
// k-steps, 8 with the four k-steps of D = 41 .. 48 (96 KiB; groups of 4 / 8 stay inside a lane).
template <int NKU, int G, int NT = 16>
__global__ __launch_bounds__(512, 2) void lnfi_kernel(
    int64_t nframes, int K, int S, int nk, const u4* __restrict__ img,
    const u4* __restrict__ Pall, float* __restrict__ log_norm, double* __restrict__ llh_sum,
    int64_t frames_per_block, const float* __restrict__ c0p) {
    constexpr int MT = 2, FW = 32, WAVES = 8, NTHREADS = 64 * WAVES;
    constexpr int kTileU4 = (NKU * NP * MT + img_nqt(NKU) * NP) * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int64_t bx;
    int by;
    {
        const int nch = (K + 16 * NT - 1) / (16 * NT);
        if (!xcd_block((nframes + frames_per_block - 1) / frames_per_block, nch, nch, bx, by))
            return;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    constexpr int p_u4 = NKU * NT * kBlockU4;
    u4* Ps = reinterpret_cast<u4*>(smem);
    {
        const u4* src = Pall + (size_t)by * nk * NT * kBlockU4;
#pragma unroll 2
        for (int idx = tid; idx < p_u4; idx += NTHREADS) Ps[idx] = src[idx];
    }
    __syncthreads();
    if (BEER_LNFI_SLEEP > 0 && wave >= WAVES / 2) {
#pragma unroll
        for (int n = 0; n < (BEER_LNFI_SLEEP + 126) / 127; ++n)
            __builtin_amdgcn_s_sleep(BEER_LNFI_SLEEP < 127 ? BEER_LNFI_SLEEP : 127);
    }
    const int kbase = by * (16 * NT);
    const float c0 = c0p[0];
    const int64_t tb = bx * frames_per_block;
    const int64_t te = tb + frames_per_block < nframes ? tb + frames_per_block : nframes;
    const u4* Pl = Ps + lane;
    u4 af[2][NP][MT];
    const int64_t fb0 = tb + (int64_t)wave * FW;
    if (fb0 < te) {
        const u4* t0 = img + (fb0 / FW) * (int64_t)kTileU4 + lane;
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int m = 0; m < MT; ++m) af[0][q][m] = t0[(q * MT + m) * 64];
    }
    for (int64_t fb = fb0; fb < te; fb += WAVES * FW) {
        const u4* ti = img + (fb / FW) * (int64_t)kTileU4 + lane;
        const bool has_next = fb + WAVES * FW < te;
        const u4* tn = ti + (has_next ? (int64_t)WAVES * kTileU4 : 0);
        f32x4 acc[MT][NT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int c = 0; c < NT; ++c) acc[m][c] = f32x4{0, 0, 0, 0};
        // B fragments: the three planes of component tile c + 1 are read from LDS at the START of
        // tile c's 12 MFMAs (hipcc otherwise places the reads behind them and waits for the
        // last one's full LDS latency before every tile: 48 stalls per wave-tile)
        u4 bp[2][NP];
#pragma unroll
        for (int pq = 0; pq < NP; ++pq) bp[0][pq] = Pl[64 * pq];
        __builtin_amdgcn_sched_group_barrier(0x100, NP, 0);
#pragma unroll
        for (int s = 0; s < NKU; ++s) {
            // the A fragments of the next k-step (of the next tile's first one: this tile's
            // again when there is none)
            {
                const u4* src = s + 1 < NKU ? ti + (size_t)(s + 1) * NP * MT * 64 : tn;
#pragma unroll
                for (int q = 0; q < NP; ++q)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        af[(s + 1) & 1][q][m] = (BEER_LNFI_ABL & 8) ? af[s & 1][q][m] : src[(q * MT + m) * 64];
            }
            if (!(BEER_LNFI_ABL & 8)) __builtin_amdgcn_sched_group_barrier(0x020, NP * MT, 0);
#pragma unroll
            for (int c = 0; c < NT; ++c) {
                constexpr int kLast = NKU * NT - 1;
                const int gi = s * NT + c, gn = gi < kLast ? gi + 1 : kLast;
#pragma unroll
                for (int pq = 0; pq < NP; ++pq)
                    bp[(gi + 1) & 1][pq] = (BEER_LNFI_ABL & 4) ? bp[gi & 1][pq] : Pl[gn * kBlockU4 + 64 * pq];
                if (BEER_LNFI_ABL & 2) continue;
#pragma unroll
                for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        acc[m][c] = mfma_bf16(af[s & 1][kProdA[pr]][m], bp[gi & 1][kProdB[pr]], acc[m][c]);
                if (!(BEER_LNFI_ABL & 4)) __builtin_amdgcn_sched_group_barrier(0x100, NP, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 6 * MT, 0);
            }
        }
        if (BEER_LNFI_ABL & 1) {
            float t = 0.f;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int c = 0; c < NT; ++c) t += acc[m][c][0] + acc[m][c][1] + acc[m][c][2] + acc[m][c][3];
            if (t == 1.2345f) log_norm[0] = t;
        } else {
            if (BEER_LNFI_ABL & 2) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int c = 0; c < NT; ++c)
                        acc[m][c] = f32x4{(float)(c + lane), (float)(m - c), (float)lane, bp[0][0][0] * 1e-30f};
            }
            lognorm_epilogue_lane_major<NT, MT, G>(acc, fb, nframes, kbase, S, i, g, lane, log_norm,
                                                   llh_sum, c0);
        }
        if constexpr (NKU % 2 == 0) {
            // (even number of k-steps: the prefetched fragments sit in af[0] already)
        } else {
#pragma unroll
            for (int q = 0; q < NP; ++q)
#pragma unroll
                for (int m = 0; m < MT; ++m) af[0][q][m] = af[1][q][m];
        }
    }
}

// component tiles per wave: 4 (64 components).  With at most 96 statistic columns
// (D <= 40) a wave's tile leaves room for two waves per SIMD: one wave's epilogue and
// fragment arithmetic run under the other's MFMAs.
inline int accf_ntc(int cov, int D) { return 4; }
inline int accf_nqt(int cov, int D) {
    const int nq = nslab_of(cov, D) * 4;
    return nq <= 96 ? 6 : (nq <= 144 ? 9 : 10);
}

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
    return supported_llh_x(D, S, S > 1 ? group_pad(G) : G);
}

inline int ntx_for(int S, int K) { return S > 1 ? 16 : (K <= 64 ? 4 : (K <= 128 ? 8 : 16)); }
inline int nchunksx_for(int S, int K) { return S > 1 ? (K + 255) / 256 : 1; }
size_t up256(size_t n) { return (n + 255) / 256 * 256; }

// bytes of the P image: nchunks x nk x NT blocks + one look-ahead batch
inline size_t p_image_bytes(int nchunks, int nk, int NT) {
    return up256(((size_t)nchunks * nk * NT + 4) * kBlockU4 * 16);
}

}  // namespace

#ifdef BEER_KERNEL_PROBE
// ISA experiments (tools/isa_stats.py): only the hot kernels, no host code
namespace {
template __global__ void llhx_kernel<16, 4, 4, true, false, false>(
    int64_t, int, int, int, int, int, int, int, const float*, const u4*, const int*, float*, float*,
    double*, float*, int, int, int, const float*, const u4*);
template __global__ void llhx_kernel<16, 2, 1, false, true, true>(
    int64_t, int, int, int, int, int, int, int, const float*, const u4*, const int*, float*, float*,
    double*, float*, int, int, int, const float*, const u4*);
template __global__ void llhx_kernel<16, 2, 1, false, true, true, true>(
    int64_t, int, int, int, int, int, int, int, const float*, const u4*, const int*, float*, float*,
    double*, float*, int, int, int, const float*, const u4*);
template __global__ void accx_kernel<false>(int64_t, int, int, int, int, const float*, const unsigned*,
                                            const int*, int64_t, double*, int, int, int,
                                            const float*, int, int);
template __global__ void accx_kernel<true>(int64_t, int, int, int, int, const float*, const unsigned*,
                                           const int*, int64_t, double*, int, int, int,
                                           const float*, int, int);
template __global__ void lnfi_kernel<3, 16>(int64_t, int, int, int, const u4*, const u4*, float*, double*,
                                            int64_t, const float*);
template __global__ void accfi_kernel<3, 8>(int64_t, int, int, int, int, int, int, const u4*, const u4*,
                                         const float*, const float*, int64_t, double*, const float*);
template __global__ void accf_kernel<4, 6, true, 8, 5, true>(int64_t, int, int, int, int, int, int, int,
                                                       const float*, const u4*, const int*,
                                                       const float*, const float*, int64_t, double*,
                                                       const float*);
}  // namespace
#else
bool supported_llh_split(int D, int S, int G) { return supported_llh_padded(D, S, G); }
// Mixture sets whose responsibilities can leave the E-step as packed tiles: full
// covariance, groups of 4 .. 128 components, a power of two
bool supported_llh_packed_sets(int cov, int D, int S, int G) {
    return cov == BEER_FULL && S > 1 && G >= 4 && G <= 128 && (G & (G - 1)) == 0 &&
           supported_llh_padded(D, S, G) && supported_acc_x(D, S * G);
}

// 129 .. 256 components of ONE mixture: the E-step kernel also leaves the
// transposed frames behind its tiles -- [R tiles][X^T tiles]
inline bool packed_has_xt(int K) { return K > kPackedComps && K <= 2 * kPackedComps; }
inline size_t packed_tiles_bytes(int64_t nframes, int K) {
    const int64_t tiles = (nframes + kPackedFrames - 1) / kPackedFrames;
    const int nblk = (K + kPackedComps - 1) / kPackedComps;
    return (size_t)tiles * nblk * NP * kPackedPlaneWords * 4;
}
size_t packed_resps_bytes(int64_t nframes, int D, int K) {
    const int64_t tiles = (nframes + kPackedFrames - 1) / kPackedFrames;
    return packed_tiles_bytes(nframes, K) +
           (packed_has_xt(K) ? (size_t)tiles * xt_pieces(D) * kPiece : 0) + 256;
}

int pack_resps(int64_t nframes, int D, int S, int G, const float* X, const float* R,
               const float* SR, void* packed, hipStream_t s) {
    const int K = S * G;
    if ((K & 3) || D < 1 || D > kMaxDimX) return BEER_EINVAL;
    if (nframes == 0) return BEER_OK;
    unsigned* tiles = reinterpret_cast<unsigned*>(packed);
    const int64_t ntile = (nframes + kPackedFrames - 1) / kPackedFrames;
    const int nblk = (K + kPackedComps - 1) / kPackedComps;
    hipLaunchKernelGGL(pack_resps_kernel, dim3((unsigned)ntile, (unsigned)nblk), dim3(256), 0, s,
                       nframes, K, S, G, R, SR, tiles);
    if (packed_has_xt(K))               // what the E-step kernel would have left behind
        hipLaunchKernelGGL(xt_image_kernel, dim3((unsigned)ntile), dim3(256), 0, s, nframes, D,
                           xt_pieces(D), X,
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

size_t estepx_workspace_bytes(int cov, int D, int S, int G) {
    if (!supported_llh_padded(D, S, G)) return 0;
    if (S > 1) G = group_pad(G);
    const int K = S * G, NT = ntx_for(S, K), nchunks = nchunksx_for(S, K);
    return p_image_bytes(nchunks, nk16_of(cov, D), NT) +
           up256((size_t)(nk16_of(cov, D) + 1) * 8 * sizeof(int)) + 256;
}

bool supported_frame_image(int cov, int D);

int estep_bf16x3(int cov, int64_t nframes, int D, int S, int G, const float* X, const float* expT,
                 const float* logw, float* resps, float* log_norm, double* llh_sum, void* ws,
                 size_t ws_bytes, hipStream_t s, bool packed, const void* image) {
    if (packed && S != 1 && !supported_llh_packed_sets(cov, D, S, G)) return BEER_EINVAL;
    if (!supported_llh_padded(D, S, G) || ws_bytes < estepx_workspace_bytes(cov, D, S, G))
        return BEER_EINVAL;
    // mixture sets whose G is not a power of two: groups padded to Gp slots (logit
    // -1e30), log-normalisers only (the responsibilities would come out in the padded
    // layout)
    const int Greal = G, Kreal = S * G;
    if (S > 1) G = group_pad(G);
    if (G != Greal && resps) return BEER_EINVAL;
    const int K = S * G;
    // a set of at most 128 single Gaussians (the per-state log-likelihoods of an HMM with one
    // Gaussian per state: 120 states at config 4) is one chunk of 8 / 4 component tiles, not
    // of 16 half of which would be padding
    const bool narrow = S > 1 && Greal == 1 && K <= 128 && !packed && !image;
    // (a frame image of four k-steps, D = 41 .. 48: chunks of 8 component tiles, whose packed
    // parameters -- 96 KiB -- fit the LDS of lnfi_kernel; groups of 4 / 8 stay inside a lane)
    const bool lnfi8 = image && !narrow && S > 1 && !resps && !packed && cov != BEER_FULL &&
                       (nslab_of(cov, D) + 7) / 8 == 4 && (G == 4 || G == 8) &&
                       beer::option(BEER_OPT_LNFI);
    const int NT = narrow ? (K <= 64 ? 4 : 8) : (lnfi8 ? 8 : ntx_for(S, K));
    const int nchunks = lnfi8 ? (K + 127) / 128 : nchunksx_for(S, K), nk = nk16_of(cov, D);
    const int kpad = nchunks * NT * 16;
    // a frame fragment image: mixture sets, log-normalisers only, groups of >= 4 (refused
    // here, before anything is launched)
    if (image && (S == 1 || packed || resps || G < 4 || cov == BEER_FULL ||
                  !supported_frame_image(cov, D)))
        return BEER_EINVAL;
    g_cov_of_launch = cov;
    char* w = reinterpret_cast<char*>(ws);
    void* P = w;
    w += p_image_bytes(nchunks, nk, NT);
    int* tab = reinterpret_cast<int*>(w);
    w += up256((size_t)(nk + 1) * 8 * sizeof(int));
    float* c0 = reinterpret_cast<float*>(w);
    hipLaunchKernelGGL(const_max_kernel, dim3(1), dim3(256), 0, s, cov, D, Kreal, expT, logw, c0);
    // log-normalisers only, groups of 4 / 8 / 16: the image is dealt out lane-major and the
    // log-sum-exp of a state stays inside a lane (lognorm_epilogue_lane_major)
    const bool lane_major = S > 1 && !resps && !packed &&
                            ((NT == 16 && (G == 4 || G == 8 || G == 16)) || lnfi8);
    hipLaunchKernelGGL(packx_kernel, dim3(kpad), dim3(256),
                       (size_t)stats_dim(cov, D) * sizeof(float), s, cov, D, Kreal, NT, expT, logw,
                       reinterpret_cast<unsigned short*>(P), tab, Greal, G, c0, lane_major ? 1 : 0);
    BEER_LAUNCH_CHECK();
    const bool full = cov == BEER_FULL;
#define BEER_LLHX(NT_, MT_, GQ_, PK_, LNO_, ...)                                                  \
    do {                                                                                          \
        if (full)                                                                                 \
            return launch_llhx<NT_, MT_, GQ_, PK_, false, LNO_>(nframes, D, K, S, G, gl, jw,      \
                                                                nchunks, nk, X, P, tab, c0,       \
                                                                resps, log_norm, llh_sum, s,      \
                                                                ##__VA_ARGS__);                   \
        return launch_llhx<NT_, MT_, GQ_, PK_, true, LNO_>(nframes, D, K, S, G, gl, jw, nchunks,  \
                                                           nk, X, P, tab, c0, resps, log_norm,    \
                                                           llh_sum, s, ##__VA_ARGS__);            \
    } while (0)
    if (S == 1) {
        const int gl = 16, jw = 4;
        if (packed) {
            if (NT == 4) BEER_LLHX(4, 2, 1, true, false);
            if (NT == 8) BEER_LLHX(8, 2, 2, true, false);
            // 129 .. 256 components: a wave's 64 frames are one tile of the accumulation;
            // it leaves them behind transposed
            float* xt = reinterpret_cast<float*>(reinterpret_cast<char*>(resps) +
                                                 packed_tiles_bytes(nframes, K));
            const int xtf = xt_pieces(D) * (kPiece / 4);
            // 32 frames x 256 components per wave, two waves per SIMD: the epilogue of one
            // wave (its 48 KB of packed tiles leave at the CU's store-issue rate) runs under
            // the other's MFMAs.  Measured at K = 256, D = 40, 1 M frames: 2.0 ms against
            // 2.3 ms for 64 x 256 per wave with one wave per SIMD (BEER_K1_WIDE=1), whose
            // hand-placed main loop runs at 90 % of the MFMA rate but whose epilogue, 0.4 ms,
            // nothing covers.
            const bool wide = beer::option(BEER_OPT_K1_WIDE) != 0;
            if (wide) BEER_LLHX(16, 4, 4, true, false, xt, xtf);
            if (full && k1_lds_fits(D, nk) && beer::option(BEER_OPT_K1_LDS))
                return launch_llhx<16, 2, 4, true, false, false, false, true>(
                    nframes, D, K, S, G, gl, jw, nchunks, nk, X, P, tab, c0, resps, log_norm,
                    llh_sum, s, xt, xtf);
            BEER_LLHX(16, 2, 4, true, false, xt, xtf);
        }
        if (NT == 4) BEER_LLHX(4, 2, 1, false, false);
        if (NT == 8) BEER_LLHX(8, 2, 2, false, false);
        BEER_LLHX(16, 2, 4, false, false);
    }
    const int jw = G < 4 ? G : 4;
    const int gl = lane_major ? 0 : (G < 4 ? 1 : (G < 64 ? G / 4 : 16));
    const int gq = G <= 64 ? 1 : G / 64;
    if (narrow) {
        if (NT == 4) BEER_LLHX(4, 2, 1, false, false);
        BEER_LLHX(8, 2, 1, false, false);
    }
    if (packed) {
        // the responsibilities within each state's mixture as the accumulation's LDS tiles
        const bool wide = beer::option(BEER_OPT_K1_WIDE) != 0;
        if (wide) {
            switch (gq) {
                case 1: BEER_LLHX(16, 4, 1, true, false);
                default: BEER_LLHX(16, 4, 2, true, false);
            }
        }
        if (full && k1_lds_fits(D, nk) && beer::option(BEER_OPT_K1_LDS)) {
            if (gq == 1)
                return launch_llhx<16, 2, 1, true, false, false, false, true>(
                    nframes, D, K, S, G, gl, jw, nchunks, nk, X, P, tab, c0, resps, log_norm,
                    llh_sum, s);
            return launch_llhx<16, 2, 2, true, false, false, false, true>(
                nframes, D, K, S, G, gl, jw, nchunks, nk, X, P, tab, c0, resps, log_norm, llh_sum,
                s);
        }
        switch (gq) {
            case 1: BEER_LLHX(16, 2, 1, true, false);
            default: BEER_LLHX(16, 2, 2, true, false);
        }
    }
    if (image && lane_major && ((nslab_of(cov, D) + 7) / 8 <= 3 || lnfi8) && beer::option(BEER_OPT_LNFI)) {
        // lane-major groups over a frame image: the chunk's parameters in LDS, a workgroup
        // walks a block of frames (lnfi_kernel)
        const int nku = (nslab_of(cov, D) + 7) / 8;
        // frames per workgroup: whole rounds of its 8 waves (256 frames); the block length
        // that minimises rounds of 256 workgroups x (block + the LDS fill, worth ~128 frames)
        int64_t best_fpb = 256, best_cost = -1;
        for (int64_t fpb = 256; fpb <= 8192; fpb += 256) {
            const int64_t wgs = (nframes + fpb - 1) / fpb * nchunks;
            const int64_t cost = (wgs + 255) / 256 * (fpb + 128);
            if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_fpb = fpb; }
        }
        const size_t lds = (size_t)nku * NT * kBlockU4 * 16;
        const int64_t gz = (nframes + best_fpb - 1) / best_fpb;
        const dim3 grid(xcd_grid(gz, nchunks, nchunks));
#define BEER_LNFI(NKU_, G_, NT_)                                                                 \
    do {                                                                                         \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lnfi_kernel<NKU_, G_, NT_>),     \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, beer::kMaxDynLds); \
        hipLaunchKernelGGL((lnfi_kernel<NKU_, G_, NT_>), grid, dim3(512), lds, s, nframes, K, S, \
                           nk, reinterpret_cast<const u4*>(image), reinterpret_cast<const u4*>(P), \
                           log_norm, llh_sum, best_fpb, c0);                                     \
    } while (0)
#define BEER_LNFI_G(NKU_)                                                                        \
    do {                                                                                         \
        if (G == 4) BEER_LNFI(NKU_, 4, 16); else if (G == 8) BEER_LNFI(NKU_, 8, 16);             \
        else BEER_LNFI(NKU_, 16, 16);                                                            \
    } while (0)
        if (lnfi8) { if (G == 4) BEER_LNFI(4, 4, 8); else BEER_LNFI(4, 8, 8); }
        else if (nku == 1) BEER_LNFI_G(1); else if (nku == 2) BEER_LNFI_G(2); else BEER_LNFI_G(3);
#undef BEER_LNFI_G
#undef BEER_LNFI
        BEER_LAUNCH_CHECK();
        return BEER_OK;
    }
    if (image) {
        // ... with the A fragments from the caller's frame fragment image
#define BEER_LNI(GQ_)                                                                            \
    return launch_llhx<16, 2, GQ_, false, true, true, true>(nframes, D, K, S, G, gl, jw, nchunks, \
                                                            nk, X, P, tab, c0, resps, log_norm,  \
                                                            llh_sum, s, nullptr, 0, image)
        switch (gq) {
            case 1: BEER_LNI(1);
            case 2: BEER_LNI(2);
            default: BEER_LNI(4);
        }
#undef BEER_LNI
    }
    if (!resps && jw == 4) {
        // log-normalisers only (the accumulation recomputes the responsibilities)
        switch (gq) {
            case 1: BEER_LLHX(16, 2, 1, false, true);
            case 2: BEER_LLHX(16, 2, 2, false, true);
            default: BEER_LLHX(16, 2, 4, false, true);
        }
    }
    switch (gq) {
        case 1: BEER_LLHX(16, 2, 1, false, false);
        case 2: BEER_LLHX(16, 2, 2, false, false);
        default: BEER_LLHX(16, 2, 4, false, false);
    }
#undef BEER_LLHX
}

size_t accx_base_workspace_bytes(int cov, int D, int K) {
    if (!supported_acc_x(D, K)) return 0;
    const int nslab = nslab_of(cov, D);
    return up256((size_t)K * nslab * 4 * sizeof(double)) + up256((size_t)nslab * sizeof(int)) + 1024;
}

size_t accx_workspace_bytes(int cov, int64_t nframes, int D, int K) {
    const size_t base = accx_base_workspace_bytes(cov, D, K);
    if (base == 0) return 0;
    const int64_t tiles = (nframes + kAxFT - 1) / kAxFT;
    return base + (size_t)tiles * xt_pieces(D) * kPiece;
}

// ... with state posteriors multiplied in by the accumulation kernel: S states of G
// components (a power of two, 8 .. 128)
bool supported_acc_sets(int cov, int D, int S, int G) {
    return S >= 1 && G >= 8 && G <= 128 && (G & (G - 1)) == 0 && supported_acc_x(D, S * G);
}
inline int acc_sets_spad(int S, int G) {
    return (S * G + kPackedComps - 1) / kPackedComps * (kPackedComps / G);
}
size_t accxs_workspace_bytes(int cov, int64_t nframes, int D, int S, int G) {
    if (!supported_acc_sets(cov, D, S, G)) return 0;
    const int64_t tiles = (nframes + kAxFT - 1) / kAxFT;
    return up256(accx_workspace_bytes(cov, nframes, D, S * G)) +
           (size_t)tiles * acc_sets_spad(S, G) * kAxFT * sizeof(float) + 1024;
}

int acc_bf16x3_packed(int cov, int64_t nframes, int D, int K, const float* X, const void* Rimg,
                      double* acc, void* ws, size_t ws_bytes, hipStream_t s, int S, int G,
                      const float* SR) {
    if (!supported_acc_x(D, K) || ws_bytes < accx_workspace_bytes(cov, nframes, D, K))
        return BEER_EINVAL;
    if (SR && (S * G != K || !supported_acc_sets(cov, D, S, G) ||
               ws_bytes < accxs_workspace_bytes(cov, nframes, D, S, G)))
        return BEER_EINVAL;
    const int nslab = nslab_of(cov, D), nq = nslab * 4;
    char* w = reinterpret_cast<char*>(ws);
    double* Sp = reinterpret_cast<double*>(w);
    w += up256((size_t)K * nq * sizeof(double));
    int* tab = reinterpret_cast<int*>(w);
    float* Xt = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) +
                                         accx_base_workspace_bytes(cov, D, K));
    hipError_t e = hipMemsetAsync(Sp, 0, (size_t)K * nq * sizeof(double), s);
    if (e != hipSuccess) return -(int)e;
    hipLaunchKernelGGL(tab_kernel, dim3(1), dim3(256), 0, s, cov, D, tab);
    BEER_LAUNCH_CHECK();
    const int64_t tiles = (nframes + kAxFT - 1) / kAxFT;
    const int NX = xt_pieces(D);
    if (packed_has_xt(K) && !SR) {           // the E-step kernel left the image behind the tiles
                                             // (one mixture; the kernel of a set does not)
        Xt = reinterpret_cast<float*>(const_cast<char*>(reinterpret_cast<const char*>(Rimg)) +
                                      packed_tiles_bytes(nframes, K));
    } else {
        hipLaunchKernelGGL(xt_image_kernel, dim3((unsigned)tiles), dim3(256), 0, s, nframes, D, NX,
                           X, Xt);
        BEER_LAUNCH_CHECK();
    }
    const int ntiles = (nq + 15) / 16;
    float* Gt = nullptr;
    int lgG = 0;
    if (SR) {
        Gt = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) +
                                      up256(accx_workspace_bytes(cov, nframes, D, K)));
        const int spad = acc_sets_spad(S, G);
        hipLaunchKernelGGL(gt_image_kernel, dim3((unsigned)tiles, (unsigned)((spad + 63) / 64)),
                           dim3(256), 0, s, nframes, S, spad, SR, Gt);
        BEER_LAUNCH_CHECK();
        while ((1 << lgG) < G) ++lgG;
    }
    const int gx = (ntiles + kAxNQ * kAxWaves - 1) / (kAxNQ * kAxWaves);
    const int gy = (K + 16 * kAxMC - 1) / (16 * kAxMC);
    // one workgroup per CU (120 KB of LDS, 512 registers per lane): whole rounds of 256
    // workgroups, at most BEER_OPT_AX_MAXFRAMES frames each
    const int chain = beer::option(BEER_OPT_AX_MAXFRAMES);
    const int64_t max_z = (nframes + 511) / 512, min_z = (nframes + chain - 1) / chain;
    const int64_t rounds = ((int64_t)gx * gy * min_z + 255) / 256;
    int64_t gz = rounds * 256 / ((int64_t)gx * gy);
    if (gz < min_z) gz = min_z;
    if (gz > max_z) gz = max_z;
    if (gz < 1) gz = 1;
    int64_t fpb = (nframes + gz - 1) / gz;
    fpb = (fpb + kAxFT - 1) / kAxFT * kAxFT;
    gz = (nframes + fpb - 1) / fpb;
    size_t lds = 2 * ((size_t)NX * kPiece + (size_t)NP * kPackedPlaneWords * 4) +
                 (SR ? 2 * 4096 : 0);
    const int sx = lds > (size_t)beer::kMaxDynLds ? 1 : 0;       // one X^T tile, two R buffers
    if (sx) lds = (size_t)NX * kPiece + 2 * (size_t)NP * kPackedPlaneWords * 4 + (SR ? 2 * 4096 : 0);
    const int64_t nyz = ((int64_t)gy * gz + 7) / 8 * 8;
    const dim3 grid((unsigned)(nyz * gx));
#define BEER_ACCX(SR_)                                                                           \
    do {                                                                                         \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(accx_kernel<SR_>),               \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, beer::kMaxDynLds);         \
        hipLaunchKernelGGL((accx_kernel<SR_>), grid, dim3(64 * kAxWaves), lds, s, nframes, D, K, \
                           nslab, NX, Xt, reinterpret_cast<const unsigned*>(Rimg), tab, fpb, Sp, \
                           gx, gy, (int)gz, Gt, lgG, sx);                                        \
    } while (0)
    if (SR) BEER_ACCX(true);
    else BEER_ACCX(false);
#undef BEER_ACCX
    BEER_LAUNCH_CHECK();
    const int64_t total = (int64_t)K * stats_dim(cov, D);
    hipLaunchKernelGGL(unpack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, cov,
                       D, K, Sp, acc);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

// statistics per Gaussian small enough for a wave's register tile: diagonal and
// isotropic covariances up to D = 64 (nq <= 160)
bool supported_accf(int cov, int D, int S, int G) {
    return cov != BEER_FULL && D >= 1 && D <= 64 && S >= 1 && G >= 1 && G <= 256 &&
           nslab_of(cov, D) * 4 <= 160;
}

size_t accf_workspace_bytes(int cov, int D, int S, int G) {
    if (!supported_accf(cov, D, S, G)) return 0;
    const int Kreal = S * G;
    G = accf_group_pad(S, G);
    const int K = S * G, NTC = accf_ntc(cov, D), nk = nk16_of(cov, D);
    const int nchunks = (K + 16 * NTC - 1) / (16 * NTC), nq = nslab_of(cov, D) * 4;
    return p_image_bytes(nchunks, nk, NTC) + up256((size_t)(nk + 1) * 8 * sizeof(int)) + 1024 +
           up256((size_t)Kreal * nq * sizeof(double));
}

// Frame fragment images (frame_image_kernel): diagonal / isotropic statistics of at most
// 96 columns in at most 3 k-steps (D <= 40; any D: the rows are staged with their padding, a
// dimension beyond D is a zero column of the tile)
bool supported_frame_image(int cov, int D) {
    // (D <= 48: at most kImgMaxNK k-steps, whose data slabs fill at most img_nqt() tiles)
    if (cov == BEER_FULL || D < 1) return false;
    const int nk_used = (nslab_of(cov, D) + 7) / 8;
    return nk_used <= kImgMaxNK && 2 * d4_of(D) <= 4 * img_nqt(nk_used);
}
size_t frame_image_bytes(int cov, int64_t nframes, int D) {
    if (!supported_frame_image(cov, D) || nframes < 0) return 0;
    const int nk = nk16_of(cov, D), nk_used = (nslab_of(cov, D) + 7) / 8;
    const int64_t tiles = (nframes + 31) / 32;
    return (size_t)tiles * frame_image_tile_u4(nk_used, img_nqt(nk_used)) * 16 +
           up256((size_t)(nk + 1) * 8 * sizeof(int));
}
__global__ void tabx_kernel(int cov, int D, int nk, int* __restrict__ tab) {
    const int Dp = 4 * d4_of(D);
    for (int s = threadIdx.x; s < (nk + 1) * 8; s += blockDim.x)
        tab[s] = s < nslab_of(cov, D) ? slab_entry(cov, D, s) : ((Dp + 2) | ((Dp + 4) << 8));
}
int frame_image(int cov, int64_t nframes, int D, const float* X, void* image, hipStream_t s) {
    if (!supported_frame_image(cov, D)) return BEER_EINVAL;
    if (nframes == 0) return BEER_OK;
    const int nk = nk16_of(cov, D), nslab = nslab_of(cov, D), nk_used = (nslab + 7) / 8;
    const int64_t tiles = (nframes + 31) / 32;
    int* tab = reinterpret_cast<int*>(reinterpret_cast<char*>(image) +
                                      (size_t)tiles * frame_image_tile_u4(nk_used, img_nqt(nk_used)) * 16);
    hipLaunchKernelGGL(tabx_kernel, dim3(1), dim3(256), 0, s, cov, D, nk, tab);
    const size_t lds = (size_t)(nk + 1) * 8 * sizeof(int) +
                       (size_t)4 * (32 * ld16_of(D) + (D + 2) * kAfXS) * sizeof(float);
    hipLaunchKernelGGL(frame_image_kernel, dim3((unsigned)((tiles + 3) / 4)), dim3(256), lds, s,
                       nframes, D, nk, nslab, img_nqt(nk_used), X, tab, reinterpret_cast<u4*>(image));
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

int acc_fused_bf16x3(int cov, int64_t nframes, int D, int S, int G, const float* X,
                     const float* expT, const float* logw, const float* log_norm,
                     const float* sr, const void* image, double* acc, void* ws, size_t ws_bytes,
                     hipStream_t s) {
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
    void* P = w;
    w += p_image_bytes(nchunks, nk, NTC);
    int* tab = reinterpret_cast<int*>(w);
    w += up256((size_t)(nk + 1) * 8 * sizeof(int));
    float* c0 = reinterpret_cast<float*>(w);
    w += 1024;
    double* Sp = reinterpret_cast<double*>(w);
    hipError_t e = hipMemsetAsync(Sp, 0, (size_t)Kreal * nq * sizeof(double), s);
    if (e != hipSuccess) return -(int)e;
    hipLaunchKernelGGL(const_max_kernel, dim3(1), dim3(256), 0, s, cov, D, Kreal, expT, logw, c0);
    hipLaunchKernelGGL(packx_kernel, dim3(kpad), dim3(256),
                       (size_t)stats_dim(cov, D) * sizeof(float), s, cov, D, Kreal, NTC, expT, logw,
                       reinterpret_cast<unsigned short*>(P), tab, Greal, G, c0, 0);
    BEER_LAUNCH_CHECK();
    // The states of every 64-component chunk fit a block of 4 consecutive ones?
    // (groups are padded to a multiple of 4: a lane's 4 components share their state)
    auto reach = [&](int ns) {
        for (int c = 0; c < nchunks; ++c) {
            const int lo = c * 16 * NTC / G, hi = (c * 16 * NTC + 60) / G;
            if ((hi < S ? hi : S - 1) - (lo < S ? lo : S - 1) > ns - 1) return false;
        }
        return true;
    };
    const bool blk = S >= 4 && reach(4);
    // ... over a frame image: a block of 4, or of 16 (groups of 4 .. 12 Gaussians)
    const int ns_img = blk ? 4 : (reach(16) ? 16 : 0);
    const int nk_used = (nslab + 7) / 8;
    // waves per workgroup: 8 (two per SIMD) with 64-component chunks, 4 with 128
    const bool use_image = image && ns_img && NTC == 4 && supported_frame_image(cov, D);
    const int waves = use_image ? beer::option(BEER_OPT_ACCFI_WAVES) : ((NTC == 4 && NQT == 6) ? 8 : 4);
    // frames per workgroup: <= kAfMaxFramesPerWave per wave, about one workgroup of
    // 8 waves (two of 4) per CU and round
    const int rounds = beer::option(BEER_OPT_ACCF_ROUNDS);
    int64_t gz = ((waves == 8 ? 256 : 512) * rounds + nchunks - 1) / nchunks;
    gz = (gz + 7) / 8 * 8;                       // whole rows of the XCD-aware grid
    // (the full chain where the launch is long enough for its last round of workgroups not
    // to matter -- 70 rounds per XCD at 10 M frames --, half of it below 4 M frames: a shard
    // of 1.25 M frames is 9 rounds per XCD with 2048-frame chains and lost 17 % to the tail)
    const int64_t chain = nframes >= ((int64_t)4 << 20) ? kAfMaxFramesPerWave : kAfMaxFramesPerWave / 2;
    const int64_t min_z = (nframes + (int64_t)waves * chain - 1) / ((int64_t)waves * chain);
    const int64_t max_z = (nframes + 32 * waves - 1) / (32 * waves);
    if (gz > max_z) gz = max_z;
    if (gz < min_z) gz = min_z;
    if (gz < 1) gz = 1;
    int64_t fpb = (nframes + gz - 1) / gz;
    fpb = (fpb + 32 * waves - 1) / (32 * waves) * (32 * waves);
    gz = (nframes + fpb - 1) / fpb;
    const size_t lds = (size_t)nk_used * NTC * kBlockU4 * 16 + (size_t)(nk + 1) * 8 * sizeof(int) +
                       (size_t)2 * NQT * 64 * sizeof(int) +
                       (size_t)waves * (32 * ld16_of(D) + (D + 2) * kAfXS + (blk ? 256 : 0)) *
                           sizeof(float);
    const dim3 grid(xcd_grid(gz, nchunks, nchunks));
    if (use_image) {
        // every fragment that depends on the frames only comes from the caller's image
        const size_t lds_i = (size_t)nk_used * NTC * kBlockU4 * 16 +
                             (size_t)waves * 64 * ns_img * sizeof(float);
        const size_t lds_red = (size_t)waves * 16 * 64 * sizeof(float);
        const size_t lds_f = lds_i > lds_red ? lds_i : lds_red;
#define BEER_ACCFI(NKU_, W_, NS_)                                                                \
    do {                                                                                         \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(accfi_kernel<NKU_, W_, NS_>),    \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, beer::kMaxDynLds); \
        hipLaunchKernelGGL((accfi_kernel<NKU_, W_, NS_>), grid, dim3(64 * W_), lds_f, s, nframes, \
                           K, S, G, Greal, nk, nslab, reinterpret_cast<const u4*>(image),        \
                           reinterpret_cast<const u4*>(P), log_norm, sr, fpb, Sp, c0);           \
    } while (0)
#define BEER_ACCFI_K(W_, NS_)                                                                    \
    do {                                                                                         \
        if (nk_used == 1) BEER_ACCFI(1, W_, NS_); else if (nk_used == 2) BEER_ACCFI(2, W_, NS_); \
        else if (nk_used == 3) BEER_ACCFI(3, W_, NS_); else BEER_ACCFI(4, W_, NS_);              \
    } while (0)
        if (waves == 8) { if (ns_img == 4) BEER_ACCFI_K(8, 4); else BEER_ACCFI_K(8, 16); }
        else { if (ns_img == 4) BEER_ACCFI_K(4, 4); else BEER_ACCFI_K(4, 16); }
#undef BEER_ACCFI_K
#undef BEER_ACCFI
        BEER_LAUNCH_CHECK();
        const int64_t total_i = (int64_t)Kreal * stats_dim(cov, D);
        hipLaunchKernelGGL(unpack_kernel, dim3((unsigned)((total_i + 255) / 256)), dim3(256), 0, s,
                           cov, D, Kreal, Sp, acc);
        BEER_LAUNCH_CHECK();
        return BEER_OK;
    }
#define BEER_ACCF(NTC_, NQT_, W_, BLK_)                                                          \
    do {                                                                                         \
        constexpr int XP_ = NQT_ == 6 ? 5 : 8;      /* D <= 40 <=> C4 <= 10 <=> nq <= 96 */       \
        (void)hipFuncSetAttribute(                                                               \
            reinterpret_cast<const void*>(accf_kernel<NTC_, NQT_, true, W_, XP_, BLK_>),         \
            hipFuncAttributeMaxDynamicSharedMemorySize, beer::kMaxDynLds);                               \
        hipLaunchKernelGGL((accf_kernel<NTC_, NQT_, true, W_, XP_, BLK_>), grid, dim3(64 * W_),  \
                           lds, s, nframes, D, K, S, G, Greal, nk, nslab, X,                     \
                           reinterpret_cast<const u4*>(P), tab, log_norm, sr, fpb, Sp, c0);      \
    } while (0)
    if (NQT == 6) { if (blk) BEER_ACCF(4, 6, 8, true); else BEER_ACCF(4, 6, 8, false); }
    else if (NQT == 9) { if (blk) BEER_ACCF(4, 9, 4, true); else BEER_ACCF(4, 9, 4, false); }
    else { if (blk) BEER_ACCF(4, 10, 4, true); else BEER_ACCF(4, 10, 4, false); }
#undef BEER_ACCF
    BEER_LAUNCH_CHECK();
    const int64_t total = (int64_t)Kreal * stats_dim(cov, D);
    hipLaunchKernelGGL(unpack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, cov,
                       D, Kreal, Sp, acc);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

#endif  // BEER_KERNEL_PROBE

}  // namespace beer_mfma
