// MFMA (matrix-core) E-step for full-covariance Gaussians on gfx950.
//
// Both halves of the E-step are GEMMs against the per-frame statistics
// phi(x) = [x, vec(x x^T), 1], which are generated on the fly from the frame
// and never stored:
//
//   K1  llh[T, K]   = PHI(X)[T, q] . P[q, K]        then row softmax (in
//                     registers) -> responsibilities + per-frame log-norm
//   K2  S[K, q]     = R^T[K, T] . PHI(X)[T, q]       gamma-weighted statistics
//
// The symmetric quadratic form is contracted over the D(D+1)/2 products
// x_a x_b, a <= b (off-diagonal coefficients doubled), i.e. about half the
// multiply-adds of the reference's dense [T, D^2+D+2] formulation.
//
// "Slab" enumeration of the contraction index (4 consecutive q = one MFMA
// k-step of v_mfma_*_16x16x4): with xe = [x_0..x_{D-1}, 0-pad to Dp = 4*D4,
// 1, 0, 0, 0] every slab is a pair (a, j) and its 4 entries are
// xe[a] * xe[4j + g], g = 0..3:
//     quadratic  a = 0..D-1, j = a/4 .. D4-1     (entries with 4j+g < a: 0)
//     linear     a = Dp ("1"), j = 0 .. D4-1
//     constant   a = Dp,      j = D4            -> (1, 0, 0, 0)
// so one uniform loop covers quadratic, linear and constant terms.
//
// Operand mapping of v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64
// (lane l: i = l & 15, g = l >> 4):  A[i][k=g], B[k=g][n=i], and C/D
// row(l, r) = 4 g + r (f32) | g + 4 r (f64), col = i.
//
// Reference restated: beer/dists/normalwishart.py:30-38, 88-92,
// beer/models/mixture.py:79-93, beer/models/normalset.py:117-123.

#include "estep_mfma.h"

#include <type_traits>

#include "common.h"
#include "estep_tiles.h"

using namespace beer;

namespace beer_mfma {

namespace {

// ---------------------------------------------------------------------------
// Parameter packing: E[T] [K, Q] (+ log weights) ->
// P[chunk][slab][64 lanes][NT] (each lane's NT B-fragment values contiguous)
// and the slab table.
// ---------------------------------------------------------------------------
// c0[0] = the largest constant term of any component: taken out of every logit (the
// accumulators round at the size of the running sum, the softmax does not see a common
// offset) and added back to the log-normalisers by the epilogue.
template <typename T>
__global__ __launch_bounds__(256) void const_max_kernel(int cov, int D, int K,
                                                        const T* __restrict__ E,
                                                        const T* __restrict__ logw,
                                                        T* __restrict__ c0) {
    __shared__ double red[8];
    const int Q = stats_dim(cov, D);
    double m = -1.0e300;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const double v = const_total(cov, D, E + (size_t)k * Q, logw ? (double)logw[k] : 0.0);
        if (v == v && v > m && v < 1.0e300) m = v;
    }
    m = block_max(m, red);
    if (threadIdx.x == 0) c0[0] = m > -1.0e300 ? (T)m : (T)0;
}

template <typename T>
__global__ void pack_kernel(int cov, int D, int K, int NT, int nchunks,
                            const T* __restrict__ E, const T* __restrict__ logw,
                            T* __restrict__ P, int* __restrict__ tab, const T* __restrict__ c0) {
    const int D4 = d4_of(D), Dp = 4 * D4, nslab = nslab_of(cov, D);
    const int Q = stats_dim(cov, D);
    const int64_t per_chunk = (int64_t)nslab_padded(cov, D) * 64 * NT;
    const int64_t total = per_chunk * nchunks;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int chunk = (int)(idx / per_chunk);
        const int64_t rem_idx = idx - (int64_t)chunk * per_chunk;
        const int c = (int)(rem_idx % NT);
        const int lane = (int)((rem_idx / NT) % 64);
        const int s = (int)(rem_idx / ((int64_t)NT * 64));
        const int i = lane & 15, g = lane >> 4;
        // column (tile c, lane-column i) of the chunk holds component
        // 64 (c / 4) + 4 i + c % 4: the four tiles of a "q-block" give every lane
        // four CONSECUTIVE components -> 16-byte stores of the responsibilities
        const int k = chunk * NT * 16 + 64 * (c >> 2) + 4 * i + (c & 3);
        const int t = slab_entry(cov, D, s);
        if (c == 0 && lane == 0 && chunk == 0) tab[s] = t;
        const int a = t & 0xff, b = ((t >> 8) & 0xff) + g, sq = t >> 16;
        double v = 0.0;
        if (s >= nslab) {
            // look-ahead padding
        } else if (k < K) {
            const T* e = E + (size_t)k * Q;
            if (sq) {                                     // diag / iso: -.5 prec_b x_b^2
                if (b < D) v = -0.5 * (double)e[cov == BEER_ISO ? D : D + b];
            } else if (a < D) {                           // full: x_a x_b, a <= b
                if (b < D && b >= a)
                    v = (b == a) ? -0.5 * (double)e[D + a * D + a]
                                 : -0.5 * ((double)e[D + a * D + b] + (double)e[D + b * D + a]);
            } else if (b - g < Dp) {                      // linear
                if (b < D) v = (double)e[b];
            } else if (g == 0) {                          // constant (its share: const_share)
                v = const_share(cov, D, s, e, logw ? (double)logw[k] : 0.0) -
                    (s == nslab - 1 ? (double)c0[0] : 0.0);
            }
        } else if (a == Dp && b - g == Dp && g == 0 && s == nslab - 1) {
            v = kPadLogit;                          // padded component: exp() -> 0
        }
        P[idx] = (T)v;
    }
}

// ---------------------------------------------------------------------------
// K1: fused log-likelihood GEMM + softmax.  One wave owns 16*MT frames and a
// chunk of 16*NT components (blockIdx.y).  Components are interleaved over the
// column tiles in "q-blocks" of 4 tiles = 64 components so that a lane holds 4
// consecutive components of a row (see pack_kernel).  The softmax runs over
// groups of G components: the whole chunk for a GMM (S = 1, GQ = NT/4
// q-blocks, gl = 16 lanes), or one state's mixture for the GMM emissions of an
// HMM (S > 1, G a power of two): GQ = G/64 q-blocks for G >= 64, else `gl` =
// G/4 lanes of one q-block (G >= 4) or `jw` = G of a lane's 4 values (G < 4).
// ---------------------------------------------------------------------------
template <typename T, int NT, int MT, int GQ>
__global__ __launch_bounds__(kThreads, (sizeof(T) == 4 && MT * NT <= 32) ? 2 : 1) void llh_kernel(
    int64_t nframes, int D, int K, int S, int G, int gl, int jw, int nslab,
    const T* __restrict__ X, const T* __restrict__ Pall, const int* __restrict__ tab,
    T* __restrict__ resps, T* __restrict__ log_norm, double* __restrict__ llh_sum,
    const T* __restrict__ c0) {
    using M = Mma<T>;
    using acc_t = typename M::acc_t;
    using vec4_t = typename M::vec4_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int D4 = d4_of(D), Dp = 4 * D4, LD = Dp + 5;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    constexpr int FW = 16 * MT;                       // frames per wave
    T* xw = reinterpret_cast<T*>(smem) + wave * (FW * LD);
    const int64_t fb = ((int64_t)blockIdx.x * (kThreads / 64) + wave) * FW;

    for (int idx = lane; idx < FW * LD; idx += 64) {
        const int r = idx / LD, c = idx - r * LD;
        const int64_t f = fb + r;
        T v = 0;
        if (c < D) { if (f < nframes) v = X[f * D + c]; }
        else if (c == Dp) v = 1;
        xw[idx] = v;
    }
    __syncthreads();

    acc_t acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int c = 0; c < NT; ++c) acc[m][c] = acc_t{0, 0, 0, 0};

    const T* xrow[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) xrow[m] = xw + (m * 16 + i) * LD;

    // Software pipeline: everything slab s+1 needs (its B fragments from the
    // packed parameter image, its two x factors from LDS, its table entry) is
    // fetched while the MFMAs of slab s run.  P and tab are padded with zero
    // slabs up to an even count + 1, so the look-ahead never branches.
    const int kbase = blockIdx.y * (16 * NT);
    const T* P = Pall + (size_t)blockIdx.y * (size_t)(nslab + 1) * 64 * NT;
    const vec4_t* Pl = reinterpret_cast<const vec4_t*>(P) + (size_t)lane * (NT / 4);
    auto fetch = [&](int s, vec4_t (&b4)[NT / 4], T (&xa)[MT], T (&xb)[MT]) {
        const int t = tab[s];
        const int jb = ((t >> 8) & 0xff) + g;
        const int a = (t >> 16) ? jb : (t & 0xff);          // square slab: xe[4j+g]^2
#pragma unroll
        for (int c4 = 0; c4 < NT / 4; ++c4) b4[c4] = Pl[(size_t)s * 64 * (NT / 4) + c4];
#pragma unroll
        for (int m = 0; m < MT; ++m) { xa[m] = xrow[m][a]; xb[m] = xrow[m][jb]; }
    };
    auto compute = [&](const vec4_t (&b4)[NT / 4], const T (&xa)[MT], const T (&xb)[MT]) {
        T av[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) av[m] = xa[m] * xb[m];
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            const T bv = b4[c / 4][c % 4];
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m][c] = M::mma(av[m], bv, acc[m][c]);
        }
    };
    vec4_t b0[NT / 4], b1[NT / 4];
    T xa0[MT], xb0[MT], xa1[MT], xb1[MT];
    fetch(0, b0, xa0, xb0);
    for (int s = 0; s < nslab; s += 2) {           // nslab_padded is even
        // sched_group_barrier: issue the look-ahead loads BEFORE the MFMA block
        // (hipcc otherwise sinks them next to their first use and exposes the
        // whole L2 latency on every slab).
        fetch(s + 1, b1, xa1, xb1);
        compute(b0, xa0, xb0);
        __builtin_amdgcn_sched_group_barrier(0x020, NT / 4, 0);     // VMEM reads
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * MT, 0);     // DS reads
        __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);    // MFMA
        fetch(s + 2, b0, xa0, xb0);
        compute(b1, xa1, xb1);
        __builtin_amdgcn_sched_group_barrier(0x020, NT / 4, 1);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * MT, 1);
        __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 1);
    }

    softmax_epilogue<T, NT, MT, GQ>(acc, fb, nframes, kbase, K, S, G, gl, jw, i, g, lane, resps,
                                    log_norm, llh_sum, c0[0]);
}

// ---------------------------------------------------------------------------
// K2: S[k, q] += sum_t r[t,k] * PHI_q(x_t).  Workgroup tile: 64 components x
// 64 slabs (256 q), 4 waves splitting the q range; grid.z splits the frames.
// Two-level fp32 accumulation (MFMA chain over kFlush frames, then a VALU
// add into a second register set) keeps the rounding error at the 1e-7
// level; the cross-workgroup reduction is fp64 atomics.
// ---------------------------------------------------------------------------
constexpr int kAccMC = 4;        // component tiles per wave (shared by the 4 waves)
constexpr int kAccFT = 64;       // frames per LDS tile
constexpr int kFlush = 256;      // frames per MFMA accumulation chain

template <typename T, int NQ, bool HAS_SR>
__global__ __launch_bounds__(kThreads, sizeof(T) == 4 ? 2 : 1) void acc_kernel(
    int64_t nframes, int D, int K, int G, int S, int nslab, const T* __restrict__ X,
    const T* __restrict__ R, const T* __restrict__ SR, const int* __restrict__ tab,
    int64_t frames_per_block, double* __restrict__ Sp) {
    using M = Mma<T>;
    using acc_t = typename M::acc_t;
    using vec4_t = typename M::vec4_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int Dp = 4 * d4_of(D);
    constexpr int RC = 16 * kAccMC;                          // components per workgroup
    // LDS, two buffers of { x tile [kAccFT][D] (raw rows), 1, 0, pad; r tile [kAccFT][RC] }
    const int xs_elems = (kAccFT * D + 2 + 3) / 4 * 4;
    const int buf_elems = xs_elems + kAccFT * RC;
    T* const lds = reinterpret_cast<T*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int kc0 = blockIdx.y * RC;
    // q tiles are dealt to the 4 waves round-robin (tile = first + 4 uu + wave) so
    // that a short statistics vector (diagonal models: 6 tiles, NQ = 2) still
    // spreads over all SIMDs.
    constexpr int kAccNQ = NQ;
    const int tile0 = blockIdx.x * (kAccNQ * (kThreads / 64)) + wave;
    const int64_t tb = (int64_t)blockIdx.z * frames_per_block;
    const int64_t te = min(nframes, tb + frames_per_block);
    const int nq = nslab * 4;

    // The two factors of this lane's statistic in each of its q tiles, as LDS
    // offsets: a real column c is relative to the frame's row (mask = ~0), the
    // constants 1 / 0 live once per buffer behind the rows (mask = 0).
    int ca[kAccNQ], cb[kAccNQ], ma[kAccNQ], mb[kAccNQ];
    const int one_off = kAccFT * D, zero_off = kAccFT * D + 1;
#pragma unroll
    for (int uu = 0; uu < kAccNQ; ++uu) {
        const int slab = 4 * (tile0 + 4 * uu) + (i >> 2);
        int a = Dp + 1, b = Dp + 1;
        if (slab < nslab) {
            const int t = tab[slab];
            b = ((t >> 8) & 0xff) + (i & 3);
            a = (t >> 16) ? b : (t & 0xff);                   // square slab
        }
        ma[uu] = a < D ? -1 : 0;
        ca[uu] = a < D ? a : (a == Dp ? one_off : zero_off);
        mb[uu] = b < D ? -1 : 0;
        cb[uu] = b < D ? b : (b == Dp ? one_off : zero_off);
    }
    // fp32: second-level accumulators (see above); fp64 needs none.
    constexpr bool kTwoLevel = sizeof(T) == 4;
    acc_t acc[kAccMC][kAccNQ], mid[kTwoLevel ? kAccMC : 1][kTwoLevel ? kAccNQ : 1];
#pragma unroll
    for (int c = 0; c < kAccMC; ++c)
#pragma unroll
        for (int uu = 0; uu < kAccNQ; ++uu) {
            acc[c][uu] = acc_t{0, 0, 0, 0};
            if (kTwoLevel) mid[c][uu] = acc_t{0, 0, 0, 0};
        }

    // Staging registers: the next tile is loaded from global memory while the
    // MFMAs of the current tile run, then written to the other LDS buffer.
    constexpr int XPT = kAccFT * (sizeof(T) == 8 ? kMaxDimF64 : kMaxDimF32) / kThreads;
    constexpr int RPT = kAccFT * RC / 4 / kThreads;                 // vec4 per thread
    const int xcount = kAccFT * D;
    T xreg[XPT];
    vec4_t rreg[RPT];
    // with G % 4 == 0 the four components of a staged vec4 share their state:
    // one state-responsibility load per vec4, its column fixed per thread
    const bool sr_vec = HAS_SR && (G % 4 == 0);
    int sr_col[HAS_SR ? RPT : 1];
    if (HAS_SR) {
#pragma unroll
        for (int v = 0; v < RPT; ++v)
            sr_col[v] = (kc0 + 4 * ((tid + v * kThreads) % (RC / 4))) / G;
    }
    auto load_tile = [&](int64_t t0) {
        const T* xsrc = X + t0 * D;
        const int64_t xvalid = (te - t0) * D;
#pragma unroll
        for (int v = 0; v < XPT; ++v) {
            const int idx = tid + v * kThreads;
            xreg[v] = (idx < xcount && idx < xvalid) ? xsrc[idx] : (T)0;
        }
#pragma unroll
        for (int v = 0; v < RPT; ++v) {
            const int e = tid + v * kThreads;
            const int r = e / (RC / 4), c4 = e % (RC / 4);
            const int64_t f = t0 + r;
            vec4_t val = vec4_t{0, 0, 0, 0};
            const int k = kc0 + 4 * c4;
            if (f < te && k + 3 < K) {
                val = *reinterpret_cast<const vec4_t*>(R + f * K + k);
                if (HAS_SR) {
                    if (sr_vec) {
                        const T w = SR[f * S + sr_col[HAS_SR ? v : 0]];
#pragma unroll
                        for (int c = 0; c < 4; ++c) val[c] *= w;
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c) val[c] *= SR[f * S + (k + c) / G];
                    }
                }
            }
            rreg[v] = val;
        }
    };
    auto store_tile = [&](int buf) {
        T* xs = lds + buf * buf_elems;
        T* rs = xs + xs_elems;
#pragma unroll
        for (int v = 0; v < XPT; ++v) {
            const int idx = tid + v * kThreads;
            if (idx < xcount) xs[idx] = xreg[v];
        }
#pragma unroll
        for (int v = 0; v < RPT; ++v)
            *reinterpret_cast<vec4_t*>(rs + (size_t)(tid + v * kThreads) * 4) = rreg[v];
    };

    if (tid < 2) {                                          // the constants, both buffers
        lds[one_off + tid] = (T)(1 - tid);
        lds[buf_elems + one_off + tid] = (T)(1 - tid);
    }
    const int64_t ntiles = (te - tb + kAccFT - 1) / kAccFT;
    if (ntiles > 0) { load_tile(tb); store_tile(0); }
    __syncthreads();
    int since_flush = 0;
    for (int64_t tile = 0; tile < ntiles; ++tile) {
        const int buf = (int)(tile & 1);
        const T* xs = lds + buf * buf_elems;
        const T* rs = xs + xs_elems;
        if (tile + 1 < ntiles) load_tile(tb + (tile + 1) * kAccFT);
#pragma unroll 2
        for (int kk = 0; kk < kAccFT / 4; ++kk) {
            // A fragments: r[frame 4kk+g][component slot 4i..4i+3] -- the 64 lanes
            // read one contiguous 1 KiB row group: conflict-free ds_read_b128.
            const vec4_t a4 = *reinterpret_cast<const vec4_t*>(rs + (4 * kk + g) * RC + 4 * i);
            const int xr = (4 * kk + g) * D;
            T bq[kAccNQ];
#pragma unroll
            for (int uu = 0; uu < kAccNQ; ++uu)
                bq[uu] = xs[(xr & ma[uu]) + ca[uu]] * xs[(xr & mb[uu]) + cb[uu]];
#pragma unroll
            for (int c = 0; c < kAccMC; ++c)
#pragma unroll
                for (int uu = 0; uu < kAccNQ; ++uu)
                    acc[c][uu] = M::mma(a4[c], bq[uu], acc[c][uu]);
        }
        since_flush += kAccFT;
        if (kTwoLevel && since_flush >= kFlush) {
            since_flush = 0;
#pragma unroll
            for (int c = 0; c < kAccMC; ++c)
#pragma unroll
                for (int uu = 0; uu < kAccNQ; ++uu) {
                    mid[c][uu] += acc[c][uu];
                    acc[c][uu] = acc_t{0, 0, 0, 0};
                }
        }
        if (tile + 1 < ntiles) store_tile(buf ^ 1);
        __syncthreads();
    }
    // rows of the C tile are component slots i' -> component kc0 + 4 i' + c
#pragma unroll
    for (int c = 0; c < kAccMC; ++c)
#pragma unroll
        for (int uu = 0; uu < kAccNQ; ++uu) {
            const int q = (tile0 + 4 * uu) * 16 + i;
            if (q >= nq) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = kc0 + 4 * M::row(g, r) + c;
                if (k < K)
                    atomicAdd(Sp + (size_t)k * nq + q,
                              kTwoLevel ? (double)(mid[c][uu][r] + acc[c][uu][r])
                                        : (double)acc[c][uu][r]);
            }
        }
}

template <typename T>
size_t align_up(size_t n) { return (n + 255) / 256 * 256; }

inline int nt_for(int S, int K) { return S > 1 ? 16 : (K <= 64 ? 4 : (K <= 128 ? 8 : 16)); }
inline int nchunks_for(int S, int K) { return S > 1 ? (K + 255) / 256 : 1; }

template <typename T, int NT, int MT, int GQ>
int launch_llh(int64_t nframes, int D, int K, int S, int G, int gl, int jw, int nchunks,
               int nslab, const T* X, const T* P, const int* tab, const T* c0, T* resps,
               T* log_norm, double* llh_sum, hipStream_t s) {
    const int D4 = d4_of(D), LD = 4 * D4 + 5;
    constexpr int FB = 16 * MT * (kThreads / 64);
    const size_t lds = (size_t)FB * LD * sizeof(T);
    const int64_t blocks = (nframes + FB - 1) / FB;
    hipLaunchKernelGGL((llh_kernel<T, NT, MT, GQ>), dim3((unsigned)blocks, (unsigned)nchunks),
                       dim3(kThreads), lds, s, nframes, D, K, S, G, gl, jw, nslab, X, P, tab,
                       resps, log_norm, llh_sum, c0);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

template <typename T>
int estep_impl(int cov, int64_t nframes, int D, int S, int G, const T* X, const T* expT,
               const T* logw, T* resps, T* log_norm, double* llh_sum, void* ws, size_t ws_bytes,
               hipStream_t s) {
    const int K = S * G;
    if (!supported_llh(D, S, G, sizeof(T)) || ws_bytes < estep_workspace_bytes(sizeof(T), cov, D, S, G))
        return BEER_EINVAL;
    const int NT = nt_for(S, K), nchunks = nchunks_for(S, K);
    const int nsp = nslab_padded(cov, D);
    const int nslab = nsp - 1;                                  // even, >= nslab_of()
    T* P = reinterpret_cast<T*>(ws);
    const size_t p_bytes = (size_t)nchunks * nsp * 64 * NT * sizeof(T);
    int* tab = reinterpret_cast<int*>(reinterpret_cast<char*>(ws) + align_up<T>(p_bytes));
    const int64_t total = (int64_t)nchunks * nsp * 64 * NT;
    int64_t pblocks = (total + 255) / 256;
    if (pblocks > 65535) pblocks = 65535;
    // (the constant's slot: behind the slab table, inside the workspace's spare 256 bytes)
    T* c0 = reinterpret_cast<T*>(reinterpret_cast<char*>(tab) +
                                 ((size_t)nsp * sizeof(int) + 15) / 16 * 16);
    hipLaunchKernelGGL(const_max_kernel<T>, dim3(1), dim3(256), 0, s, cov, D, K, expT, logw, c0);
    hipLaunchKernelGGL(pack_kernel<T>, dim3((unsigned)pblocks), dim3(256), 0, s, cov, D, K, NT,
                       nchunks, expT, logw, P, tab, c0);
    BEER_LAUNCH_CHECK();
    constexpr int MT = sizeof(T) == 4 ? 2 : 1;
#define BEER_LLH(NT_, GQ_) \
    return launch_llh<T, NT_, MT, GQ_>(nframes, D, K, S, G, gl, jw, nchunks, nslab, X, P, tab, \
                                       c0, resps, log_norm, llh_sum, s)
    if (S == 1) {                                   // one group = the whole (padded) chunk
        const int gl = 16, jw = 4;
        if (NT == 4) BEER_LLH(4, 1);
        if (NT == 8) BEER_LLH(8, 2);
        BEER_LLH(16, 4);
    }
    const int jw = G < 4 ? G : 4;
    const int gl = G < 4 ? 1 : (G < 64 ? G / 4 : 16);
    const int gq = G <= 64 ? 1 : G / 64;
    switch (gq) {
        case 1: BEER_LLH(16, 1);
        case 2: BEER_LLH(16, 2);
        default: BEER_LLH(16, 4);
    }
#undef BEER_LLH
}

template <typename T>
int acc_impl(int cov, int64_t nframes, int D, int S, int G, const T* X, const T* R, const T* SR,
             double* acc, void* ws, size_t ws_bytes, hipStream_t s) {
    const int K = S * G;
    if (!supported_acc(D, K, sizeof(T)) || ws_bytes < acc_workspace_bytes(cov, D, K, sizeof(T)))
        return BEER_EINVAL;
    const int nslab = nslab_of(cov, D), nq = nslab * 4;
    double* Sp = reinterpret_cast<double*>(ws);
    int* tab = reinterpret_cast<int*>(reinterpret_cast<char*>(ws) +
                                      align_up<T>((size_t)K * nq * sizeof(double)));
    hipLaunchKernelGGL(tab_kernel, dim3(1), dim3(256), 0, s, cov, D, tab);
    BEER_LAUNCH_CHECK();
    hipError_t e = hipMemsetAsync(Sp, 0, (size_t)K * nq * sizeof(double), s);
    if (e != hipSuccess) return -(int)e;
    const int ntiles = (nq + 15) / 16;
    const int NQ = ntiles > 8 ? 4 : (ntiles > 4 ? 2 : 1);       // q tiles per wave
    const int gx = (ntiles + NQ * (kThreads / 64) - 1) / (NQ * (kThreads / 64));
    const int gy = (K + 16 * kAccMC - 1) / (16 * kAccMC);
    int64_t gz = (1024 + (int64_t)gx * gy - 1) / ((int64_t)gx * gy);
    const int64_t max_z = (nframes + 1023) / 1024;
    if (gz > max_z) gz = max_z;
    if (gz < 1) gz = 1;
    int64_t fpb = (nframes + gz - 1) / gz;
    fpb = (fpb + kAccFT - 1) / kAccFT * kAccFT;
    gz = (nframes + fpb - 1) / fpb;
    const size_t lds = 2 * ((size_t)(kAccFT * D + 2 + 3) / 4 * 4 + (size_t)kAccFT * 16 * kAccMC) *
                       sizeof(T);
    const dim3 grid(gx, gy, (unsigned)gz);
#define BEER_ACC(NQ_, SR_)                                                                      \
    do {                                                                                        \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(acc_kernel<T, NQ_, SR_>),       \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, beer::kMaxDynLds);        \
        hipLaunchKernelGGL((acc_kernel<T, NQ_, SR_>), grid, dim3(kThreads), lds, s, nframes, D, \
                           K, G, S, nslab, X, R, SR, tab, fpb, Sp);                             \
    } while (0)
    if (SR) {
        if (NQ == 4) BEER_ACC(4, true);
        else if (NQ == 2) BEER_ACC(2, true);
        else BEER_ACC(1, true);
    } else {
        if (NQ == 4) BEER_ACC(4, false);
        else if (NQ == 2) BEER_ACC(2, false);
        else BEER_ACC(1, false);
    }
#undef BEER_ACC
    BEER_LAUNCH_CHECK();
    const int64_t total = (int64_t)K * stats_dim(cov, D);
    hipLaunchKernelGGL(unpack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, cov,
                       D, K, Sp, acc);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

}  // namespace

static bool supported_llh_dim(int D, int S, int G, int max_d) {
    // 8-bit slab-table fields: Dp + 4 <= 255.  GMM (S = 1): any K in [16, 256]
    // (a whole softmax row inside one wave's accumulators, padded to 64 / 128 /
    // 256 columns).  Mixture set (S > 1): G a power of two <= 256 so that the
    // groups align with lanes / column tiles; K is cut into chunks of 256.
    if (D < 1 || D > max_d || S < 1 || G < 1) return false;
    const int K = S * G;
    if (S == 1) return K >= 16 && K <= 256;
    return K >= 16 && G <= 256 && (G & (G - 1)) == 0;
}
bool supported_llh(int D, int S, int G, size_t elem) { return supported_llh_dim(D, S, G, max_dim(elem)); }
bool supported_llh_x(int D, int S, int G) { return supported_llh_dim(D, S, G, kMaxDimX); }

bool supported_acc(int D, int K, size_t elem) {
    return D >= 1 && D <= max_dim(elem) && K >= 16 && K % 4 == 0;
}
bool supported_acc_x(int D, int K) { return D >= 1 && D <= kMaxDimX && K >= 16 && K % 4 == 0; }

size_t estep_workspace_bytes(size_t elem, int cov, int D, int S, int G) {
    if (!supported_llh(D, S, G, elem)) return 0;
    const int nslab = nslab_padded(cov, D), K = S * G;
    return (size_t)(((size_t)nchunks_for(S, K) * nslab * 64 * nt_for(S, K) * elem + 255) / 256 *
                    256) +
           (size_t)nslab * sizeof(int) + 256;
}

size_t acc_workspace_bytes(int cov, int D, int K, size_t elem) {
    if (!supported_acc(D, K, elem)) return 0;
    const int nslab = nslab_of(cov, D);
    return (size_t)(((size_t)K * nslab * 4 * sizeof(double) + 255) / 256 * 256) +
           (size_t)nslab * sizeof(int) + 256;
}

int estep_f32(int cov, int64_t T, int D, int S, int G, const float* X, const float* expT,
              const float* logw, float* resps, float* log_norm, double* llh_sum, void* ws,
              size_t ws_bytes, hipStream_t s) {
    return estep_impl<float>(cov, T, D, S, G, X, expT, logw, resps, log_norm, llh_sum, ws,
                             ws_bytes, s);
}
int estep_f64(int cov, int64_t T, int D, int S, int G, const double* X, const double* expT,
              const double* logw, double* resps, double* log_norm, double* llh_sum, void* ws,
              size_t ws_bytes, hipStream_t s) {
    return estep_impl<double>(cov, T, D, S, G, X, expT, logw, resps, log_norm, llh_sum, ws,
                              ws_bytes, s);
}
int acc_f32(int cov, int64_t T, int D, int S, int G, const float* X, const float* R,
            const float* SR, double* acc, void* ws, size_t ws_bytes, hipStream_t s) {
    return acc_impl<float>(cov, T, D, S, G, X, R, SR, acc, ws, ws_bytes, s);
}
int acc_f64(int cov, int64_t T, int D, int S, int G, const double* X, const double* R,
            const double* SR, double* acc, void* ws, size_t ws_bytes, hipStream_t s) {
    return acc_impl<double>(cov, T, D, S, G, X, R, SR, acc, ws, ws_bytes, s);
}

}  // namespace beer_mfma
