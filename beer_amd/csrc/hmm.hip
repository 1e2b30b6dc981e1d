// HMM inference over a ragged batch of utterances: pdf-id gather / scatter,
// log-space forward-backward with per-frame-normalised state posteriors and
// summed transition posteriors, Viterbi with backtrack, one-hot posteriors of
// a given path.
//
// One workgroup per utterance (the recursions are sequential in time; the
// parallelism is across utterances and across states), the previous column of
// the trellis lives in LDS, transitions are CSR (the reference's dense [S,S]
// log-matrix with its -inf entries dropped -- exp(-inf) contributes exactly 0).
// Forward-backward arithmetic is fp64 whatever the storage type; Viterbi
// arithmetic is done in the storage type because its int64 output must be
// bit-identical to the reference's recursion (one add per hypothesis).
//
// Reference restated: beer/graph.py:270-344, beer/models/hmm.py:40-121,
// beer/models/modelset.py:140-154.

#include "common.h"

using namespace beer;

namespace {

constexpr int kHmmThreads = 256;

template <typename T>
__device__ __forceinline__ T ninf() { return (T)-INFINITY; }

// exp / log of the forward-backward recursions in the model's own precision
// (an fp64 exp costs ~8x an fp32 one on the VALU and the recursion is
// transcendental-bound); maxima, sums and the trellis itself stay fp64.
template <typename T> __device__ __forceinline__ double fexp(double x) { return (double)exp((T)x); }
template <typename T> __device__ __forceinline__ double flog(double x) { return (double)log((T)x); }
// float: the hardware's 2^x and log2 (1 ulp) around an fp64 change of base -- the
// argument is rounded to float once, as in expf((float)x), at a quarter of the
// instructions.  Arguments are <= 0 after the max subtraction (results below
// 2^-126 flush to 0: they are added to a term equal to 1), logs are taken of
// sums >= 1.
template <> __device__ __forceinline__ double fexp<float>(double x) {
    return (double)__builtin_amdgcn_exp2f((float)(x * 1.4426950408889634074));
}
template <> __device__ __forceinline__ double flog<float>(double x) {
    return (double)__builtin_amdgcn_logf((float)x) * 0.69314718055994530942;
}

// ---------------------------------------------------------------------------
// gather: pc_llhs[u][t,s] = scale * pc_all[frame_off[u]+t, pdf_ids[...]]
// ---------------------------------------------------------------------------
template <typename T>
__global__ void gather_kernel(beer_batch b, int S_total, const T* __restrict__ pc_all, T scale,
                              T* __restrict__ out) {
    const int u = blockIdx.x;
    const int gid = b.graph_id[u];
    const int S = b.graphs[gid].n_states;
    const int32_t* ids = b.pdf_ids + b.pdf_off[gid];
    const int64_t f0 = b.frame_off[u], nt = b.frame_off[u + 1] - f0;
    T* o = out + b.llh_off[u];
    for (int64_t idx = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; idx < nt * S;
         idx += (int64_t)gridDim.y * blockDim.x) {
        const int64_t t = idx / S;
        const int s = (int)(idx % S);
        o[idx] = scale * pc_all[(f0 + t) * S_total + ids[s]];
    }
}

// scatter (+ per-frame expected llh): one wave per frame, lanes over the
// utterance's states (coalesced rows of gamma / pc); repeated pdf ids
// (alignment graphs) make the add atomic.
template <typename T>
__global__ void scatter_kernel(beer_batch b, int S_total, const T* __restrict__ pc,
                               const T* __restrict__ gamma, T scale, T* __restrict__ sr,
                               T* __restrict__ exp_llh, double* __restrict__ utt_llh) {
    __shared__ double red[8];
    const int u = blockIdx.x;
    const int gid = b.graph_id[u];
    const int S = b.graphs[gid].n_states;
    const int32_t* ids = b.pdf_ids + b.pdf_off[gid];
    const int64_t f0 = b.frame_off[u], nt = b.frame_off[u + 1] - f0;
    const T* g = gamma + b.llh_off[u];
    const T* p = pc + b.llh_off[u];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    double mine = 0.0;
    for (int64_t t = (int64_t)blockIdx.y * nwave + wave; t < nt;
         t += (int64_t)gridDim.y * nwave) {
        T e = 0;
        T* row = sr ? sr + (f0 + t) * S_total : nullptr;
        for (int s = lane; s < S; s += 64) {
            const T gv = g[t * S + s];
            if (row) atomicAdd(row + ids[s], scale * gv);
            e += p[t * S + s] * gv;
        }
        e = wave_sum(e);
        if (lane == 0) {
            if (exp_llh) exp_llh[f0 + t] = e;
            mine += (double)e;
        }
    }
    if (utt_llh) {
        const double tot = block_sum(mine, red);
        if (threadIdx.x == 0) atomicAdd(utt_llh + u, tot);
    }
}

// Element-parallel scatter for the batched path (no per-frame exp_llh wanted):
// coalesced reads of gamma / pc, per-utterance sum of gamma * pc reduced per
// workgroup.  Repeated pdf ids (alignment graphs) need the atomic add.
template <typename T>
__global__ void scatter_flat_kernel(beer_batch b, int S_total, const T* __restrict__ pc,
                                    const T* __restrict__ gamma, T scale, T* __restrict__ sr,
                                    double* __restrict__ utt_llh) {
    __shared__ double red[8];
    const int u = blockIdx.x;
    const int gid = b.graph_id[u];
    const int S = b.graphs[gid].n_states;
    const int32_t* ids = b.pdf_ids + b.pdf_off[gid];
    const int64_t f0 = b.frame_off[u], nt = b.frame_off[u + 1] - f0;
    const T* g = gamma + b.llh_off[u];
    const T* p = pc + b.llh_off[u];
    double mine = 0.0;
    for (int64_t idx = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; idx < nt * S;
         idx += (int64_t)gridDim.y * blockDim.x) {
        const int64_t t = idx / S;
        const int s = (int)(idx - t * S);
        const T gv = g[idx];
        if (sr) atomicAdd(sr + (f0 + t) * S_total + ids[s], scale * gv);
        mine += (double)(p[idx] * gv);
    }
    if (utt_llh) {
        const double tot = block_sum(mine, red);
        if (threadIdx.x == 0) atomicAdd(utt_llh + u, tot);
    }
}

// ---------------------------------------------------------------------------
// forward-backward
// ---------------------------------------------------------------------------
// The recursions are parallel over ARCS and row SEGMENTS, not states: a
// phone-loop graph has a few states with ~P incoming (phone starts) or outgoing
// (phone ends) arcs, and a thread-per-state loop would serialise P dependent
// steps behind one lane.  Each log-sum-exp is five short phases over LDS:
// partial max per segment (<= BEER_SEG arcs), max per state, p_e = exp(v_e - max)
// per arc, partial sum per segment, sum + log per state.  The graph (CSR
// indices as int16, weights) is copied into LDS once per utterance, so the
// T sequential steps never touch global memory for the topology.  The
// transition posteriors reuse the backward pass's p_e:
//   xi_t(i,j) = p_e * gamma_t(i) / s_i,   s_i = sum_e p_e,
// and cost no transcendental at all.
constexpr int kFbThreads = 512;

struct FbLayout {            // byte offsets into dynamic LDS, from the batch maxima
    size_t cur, nxt, lb, mx, red, v, xi, pseg, w_in, w_out, src_in, dst_in, dst_out, src_out,
        seg_in, rseg_in, seg_out, rseg_out, total;
    __host__ __device__ FbLayout(int S, int A, int G, bool has_xi, size_t elem) {
        size_t o = 0;
        auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 15) / 16 * 16; return r; };
        cur = take(8 * (size_t)S); nxt = take(8 * (size_t)S); lb = take(8 * (size_t)S);
        mx = take(8 * (size_t)S); red = take(8 * 16);
        v = take(8 * (size_t)A); xi = take(has_xi ? 8 * (size_t)A : 0); pseg = take(8 * (size_t)G);
        w_in = take(elem * A); w_out = take(elem * A);
        src_in = take(2 * (size_t)A); dst_in = take(2 * (size_t)A);
        dst_out = take(2 * (size_t)A); src_out = take(2 * (size_t)A);
        seg_in = take(4 * (size_t)(G + 1)); rseg_in = take(4 * (size_t)(S + 1));
        seg_out = take(4 * (size_t)(G + 1)); rseg_out = take(4 * (size_t)(S + 1));
        total = o;
    }
};

// BIG: a graph whose arc lists do not fit a CU's LDS (a dense 100-state HMM has
// 10 000 arcs).  Same phases; the topology is read where it lies (the graph image in
// global memory: L2 resident, int32 indices), the per-arc scratch (p_e, xi) and the
// per-segment partials live in a slice of `arc_ws` per workgroup (2 max_arcs +
// max_segs doubles), only the per-state arrays stay in LDS, and a bounded number of
// workgroups walks the utterances (so that the scratch is bounded too).
constexpr size_t kLdsBytes = 160 * 1024;
inline size_t fb_lds_bytes(const beer_batch* b, int dtype, bool has_xi, bool big) {
    return FbLayout(b->max_states, big ? 0 : b->max_arcs, big ? 0 : b->max_segs, !big && has_xi,
                    dtype == BEER_F32 ? 4 : 8).total;
}
template <bool BIG> struct FbIdx { typedef int16_t type; };
template <> struct FbIdx<true> { typedef int32_t type; };
constexpr int kFbBigBlocks = 512;

template <typename T, bool BIG>
__device__ __forceinline__ void fb_utterance(
    const beer_batch& b, int u, const T* __restrict__ pc_llhs, double* __restrict__ alpha_ws,
    T* __restrict__ gamma, double* __restrict__ xi_sum, double* __restrict__ gamma0_sum,
    T* __restrict__ lognorm_mean, double* __restrict__ arc_ws, char* smem) {
    typedef typename FbIdx<BIG>::type idx_t;
    const int tid = threadIdx.x, nt_ = blockDim.x;
    const beer_graph g = b.graphs[b.graph_id[u]];
    const int S = g.n_states, nnz = g.n_arcs, nsi = g.n_in_seg, nso = g.n_out_seg;
    const int64_t T_ = b.frame_off[u + 1] - b.frame_off[u];
    if (T_ <= 0) return;
    const T* llh = pc_llhs + b.llh_off[u];
    double* alpha = alpha_ws + b.llh_off[u];
    T* gam = gamma + b.llh_off[u];
    const T* init = (const T*)g.init;
    const T* fin = (const T*)g.final;

    const FbLayout L(b.max_states, BIG ? 0 : b.max_arcs, BIG ? 0 : b.max_segs,
                     !BIG && xi_sum != nullptr, sizeof(T));
    double* cur = reinterpret_cast<double*>(smem + L.cur);      // alpha_{t-1} / beta_t
    double* nxt = reinterpret_cast<double*>(smem + L.nxt);      // alpha_t / s_i / gamma_t/s_i
    double* lb = reinterpret_cast<double*>(smem + L.lb);        // llh_{t+1} + beta_{t+1}
    double* mx = reinterpret_cast<double*>(smem + L.mx);        // per-state max
    double* red = reinterpret_cast<double*>(smem + L.red);
    int* rseg_in = reinterpret_cast<int*>(smem + L.rseg_in);
    int* rseg_out = reinterpret_cast<int*>(smem + L.rseg_out);
    double *v, *xi, *pseg;                  // per-arc scratch, per-segment partials
    const T *w_in, *w_out;
    const idx_t *src_in, *dst_in, *dst_out, *src_out;
    const int *seg_in, *seg_out;
    if constexpr (BIG) {
        double* mine = arc_ws + (size_t)blockIdx.x * (2 * (size_t)b.max_arcs + b.max_segs);
        v = mine; xi = mine + b.max_arcs; pseg = mine + 2 * (size_t)b.max_arcs;
        w_in = (const T*)g.in_w; w_out = (const T*)g.out_w;
        src_in = g.in_src; dst_in = g.in_dst; dst_out = g.out_dst; src_out = g.out_src;
        seg_in = g.in_seg; seg_out = g.out_seg;
        if (xi_sum)
            for (int e = tid; e < nnz; e += nt_) xi[e] = 0.0;
    } else {
        // ---- topology -> LDS ----
        v = reinterpret_cast<double*>(smem + L.v);
        xi = reinterpret_cast<double*>(smem + L.xi);
        pseg = reinterpret_cast<double*>(smem + L.pseg);
        T* wi = reinterpret_cast<T*>(smem + L.w_in);
        T* wo = reinterpret_cast<T*>(smem + L.w_out);
        int16_t* si = reinterpret_cast<int16_t*>(smem + L.src_in);
        int16_t* di = reinterpret_cast<int16_t*>(smem + L.dst_in);
        int16_t* dO = reinterpret_cast<int16_t*>(smem + L.dst_out);
        int16_t* sO = reinterpret_cast<int16_t*>(smem + L.src_out);
        int* sgi = reinterpret_cast<int*>(smem + L.seg_in);
        int* sgo = reinterpret_cast<int*>(smem + L.seg_out);
        for (int e = tid; e < nnz; e += nt_) {
            wi[e] = ((const T*)g.in_w)[e];
            wo[e] = ((const T*)g.out_w)[e];
            si[e] = (int16_t)g.in_src[e];
            di[e] = (int16_t)g.in_dst[e];
            dO[e] = (int16_t)g.out_dst[e];
            sO[e] = (int16_t)g.out_src[e];
            if (xi_sum) xi[e] = 0.0;
        }
        for (int k = tid; k <= nsi; k += nt_) sgi[k] = g.in_seg[k];
        for (int k = tid; k <= nso; k += nt_) sgo[k] = g.out_seg[k];
        w_in = wi; w_out = wo; src_in = si; dst_in = di; dst_out = dO; src_out = sO;
        seg_in = sgi; seg_out = sgo;
    }
    const double NINF = neg_inf(), PINF = __builtin_huge_val();
    for (int k = tid; k <= S; k += nt_) { rseg_in[k] = g.in_row_seg[k]; rseg_out[k] = g.out_row_seg[k]; }

    // ---- forward ----
    for (int j = tid; j < S; j += nt_) {
        const double a = (double)llh[j] + (double)init[j];
        cur[j] = a;
        alpha[j] = a;
    }
    __syncthreads();
    for (int64_t t = 1; t < T_; ++t) {
        for (int k = tid; k < nsi; k += nt_) {                // partial max per segment
            double m = NINF;
            for (int e = seg_in[k]; e < seg_in[k + 1]; ++e) {
                const double w = cur[src_in[e]] + (double)w_in[e];
                m = w > m ? w : m;
            }
            pseg[k] = m;
        }
        __syncthreads();
        for (int j = tid; j < S; j += nt_) {                  // max per state
            double m = NINF;
            for (int k = rseg_in[j]; k < rseg_in[j + 1]; ++k) m = pseg[k] > m ? pseg[k] : m;
            mx[j] = m;
        }
        __syncthreads();
        for (int e = tid; e < nnz; e += nt_) {                // exp per arc
            const double m = mx[dst_in[e]];
            v[e] = (m > NINF && m < PINF) ? fexp<T>(cur[src_in[e]] + (double)w_in[e] - m) : 0.0;
        }
        __syncthreads();
        for (int k = tid; k < nsi; k += nt_) {                // partial sum per segment
            double sm = 0.0;
            for (int e = seg_in[k]; e < seg_in[k + 1]; ++e) sm += v[e];
            pseg[k] = sm;
        }
        __syncthreads();
        for (int j = tid; j < S; j += nt_) {                  // sum + log per state
            const double m = mx[j];
            double lse = m;
            if (m > NINF && m < PINF) {
                double sm = 0.0;
                for (int k = rseg_in[j]; k < rseg_in[j + 1]; ++k) sm += pseg[k];
                lse = m + flog<T>(sm);
            }
            const double a = (double)llh[t * S + j] + lse;
            nxt[j] = a;
            alpha[t * S + j] = a;
        }
        __syncthreads();
        double* tmp = cur; cur = nxt; nxt = tmp;
    }

    // ---- backward + posteriors ----
    for (int j = tid; j < S; j += nt_) cur[j] = (double)fin[j];      // beta_{T-1}
    __syncthreads();
    double ln_acc = 0.0;
    for (int64_t t = T_ - 1; t >= 0; --t) {
        const bool inner = t < T_ - 1;
        if (inner) {
            // beta_t(i) = lse_j(A_ij + llh_{t+1}(j) + beta_{t+1}(j)); keep p_e, s_i
            for (int k = tid; k < nso; k += nt_) {
                double m = NINF;
                for (int e = seg_out[k]; e < seg_out[k + 1]; ++e) {
                    const double w = (double)w_out[e] + lb[dst_out[e]];
                    m = w > m ? w : m;
                }
                pseg[k] = m;
            }
            __syncthreads();
            for (int i = tid; i < S; i += nt_) {
                double m = NINF;
                for (int k = rseg_out[i]; k < rseg_out[i + 1]; ++k) m = pseg[k] > m ? pseg[k] : m;
                mx[i] = m;
            }
            __syncthreads();
            for (int e = tid; e < nnz; e += nt_) {
                const double m = mx[src_out[e]];
                v[e] = (m > NINF && m < PINF)
                           ? fexp<T>((double)w_out[e] + lb[dst_out[e]] - m) : 0.0;
            }
            __syncthreads();
            for (int k = tid; k < nso; k += nt_) {
                double sm = 0.0;
                for (int e = seg_out[k]; e < seg_out[k + 1]; ++e) sm += v[e];
                pseg[k] = sm;
            }
            __syncthreads();
            for (int i = tid; i < S; i += nt_) {
                const double m = mx[i];
                double sm = 0.0, lse = m;
                if (m > NINF && m < PINF) {
                    for (int k = rseg_out[i]; k < rseg_out[i + 1]; ++k) sm += pseg[k];
                    lse = m + flog<T>(sm);
                }
                cur[i] = lse;
                nxt[i] = sm;                                 // s_i, replaced by gamma/s below
            }
            __syncthreads();
        }
        // lognorm_t = lse_i(alpha_t(i) + beta_t(i)); gamma_t
        double m = NINF;
        for (int j = tid; j < S; j += nt_) {
            const double w = alpha[t * S + j] + cur[j];
            m = w > m ? w : m;
        }
        m = block_max(m, red);
        double lognorm = m;
        double mine = 0.0;
        const bool finite = m > NINF && m < PINF;
        if (finite) {
            for (int j = tid; j < S; j += nt_) mine += fexp<T>(alpha[t * S + j] + cur[j] - m);
            const double sm = block_sum(mine, red);
            lognorm = m + flog<T>(sm);
        }
        ln_acc += lognorm;
        for (int j = tid; j < S; j += nt_) {
            // NaN when alpha + beta and lognorm are both -inf, as in the reference
            const double gv = fexp<T>(alpha[t * S + j] + cur[j] - lognorm);
            gam[t * S + j] = (T)gv;
            if (t == 0 && gamma0_sum) atomicAdd(gamma0_sum + j, gv);
            if (inner) {
                const double sm = nxt[j];
                nxt[j] = (sm > 0.0 && gv == gv) ? gv / sm : 0.0;           // NaN -> 0
            }
            lb[j] = (double)llh[t * S + j] + cur[j];                      // for frame t-1
        }
        __syncthreads();
        // xi_t(i,j) for the arcs t -> t+1 (graph.py:308-323)
        if (xi_sum && inner && lognorm > NINF) {
            for (int e = tid; e < nnz; e += nt_) xi[e] += v[e] * nxt[src_out[e]];
        }
        // (v / nxt are next written after barriers of the next frame's backward
        //  step: no hazard with the xi loop)
    }
    if (lognorm_mean && tid == 0) lognorm_mean[u] = (T)(ln_acc / (double)T_);
    if (xi_sum) {
        __syncthreads();
        for (int e = tid; e < nnz; e += nt_)
            atomicAdd(xi_sum + (size_t)src_out[e] * S + dst_out[e], xi[e]);
    }
}

template <typename T, bool BIG>
__global__ __launch_bounds__(kFbThreads) void fb_kernel(
    beer_batch b, const T* __restrict__ pc_llhs, double* __restrict__ alpha_ws,
    T* __restrict__ gamma, double* __restrict__ xi_sum, double* __restrict__ gamma0_sum,
    T* __restrict__ lognorm_mean, double* __restrict__ arc_ws) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int u = blockIdx.x; u < b.nutt; u += gridDim.x) {       // (one pass unless BIG)
        fb_utterance<T, BIG>(b, u, pc_llhs, alpha_ws, gamma, xi_sum, gamma0_sum, lognorm_mean,
                             arc_ws, smem);
        __syncthreads();                                         // LDS reused by the next one
    }
}

// ---------------------------------------------------------------------------
// forward-backward, low-degree + hub variant (see beer_graph_lowdeg).  Every
// state has <= BEER_SEG sparse arcs, so one thread per state finishes its
// log-sum-exp alone; a hub (the eliminated pivot of a phone loop) is one
// log-sum-exp over <= a few dozen sources done by wave 0 with shuffles while
// the other waves work on the sparse arcs.  Barriers per frame: 2 forward,
// 4 backward (the general kernel above needs 5 + 9).
// ---------------------------------------------------------------------------
constexpr int kLdThreads = 512;     // one state per thread: graphs up to 512 states
constexpr int kMaxHubs = 4;
constexpr int kRegHubs = 1;         // hubs whose member lists are cached in registers

template <typename T>
__device__ __forceinline__ double logaddexp2(double a, double b) {
    const double m = a > b ? a : b;
    if (!(m > neg_inf())) return m;                       // both -inf
    return m + flog<T>(fexp<T>(a - m) + fexp<T>(b - m));
}

template <typename T>
__global__ __launch_bounds__(kLdThreads) void fb_lowdeg_kernel(
    beer_batch b, const T* __restrict__ pc_llhs, double* __restrict__ alpha_ws,
    T* __restrict__ gamma, double* __restrict__ xi_sum, double* __restrict__ gamma0_sum,
    double* __restrict__ hub_flow, T* __restrict__ lognorm_mean) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int u = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const beer_graph g = b.graphs[b.graph_id[u]];
    const beer_graph_lowdeg L = *g.lowdeg;
    const int S = g.n_states, H = L.n_hubs;
    const int64_t T_ = b.frame_off[u + 1] - b.frame_off[u];
    if (T_ <= 0) return;
    const T* llh = pc_llhs + b.llh_off[u];
    double* alpha = alpha_ws + b.llh_off[u];
    T* gam = gamma + b.llh_off[u];
    const T* in_w = (const T*)L.in_w;
    const T* out_w = (const T*)L.out_w;
    const T* hsw = (const T*)L.hub_src_w;
    const T* hdw = (const T*)L.hub_dst_w;
    const double NINF = neg_inf(), PINF = __builtin_huge_val();

    double* cur = reinterpret_cast<double*>(smem);       // [S] alpha_{t-1} / beta_t
    double* lb = cur + b.max_states;                     // [S] llh_{t+1} + beta_{t+1}
    // hub[0..kMaxHubs): hub value of the running recursion; hub[kMaxHubs]:
    // lognorm_t; hub[kMaxHubs+1 ..): forward hub values H_h(t), recomputed in the
    // backward pass from alpha_t by wave 1 (needed for the hub flows)
    double* al = lb + b.max_states;                      // [S] alpha_t of the backward frame
    double* hub = al + b.max_states;

    // One state per thread (S <= blockDim is required by the launcher).
    const int j = tid;
    const bool st = j < S;
    // this state's sparse arcs, weights and hub links in registers
    int in_beg = 0, in_end = 0, out_beg = 0, out_end = 0, hs = -1, hd = -1;
    double hs_w = 0.0, hd_w = 0.0;
    if (st) {
        in_beg = L.in_ptr[j]; in_end = L.in_ptr[j + 1];
        out_beg = L.out_ptr[j]; out_end = L.out_ptr[j + 1];
        hs = L.hub_src_id[j]; hd = L.hub_dst_id[j];
        if (hs >= 0) hs_w = (double)hsw[j];
        if (hd >= 0) hd_w = (double)hdw[j];
    }
    int isrc[BEER_SEG], odst[BEER_SEG];
    double iw[BEER_SEG], ow[BEER_SEG], xi_r[BEER_SEG];
#pragma unroll
    for (int k = 0; k < BEER_SEG; ++k) {
        const bool a = in_beg + k < in_end, o = out_beg + k < out_end;
        isrc[k] = a ? L.in_src[in_beg + k] : 0;
        iw[k] = a ? (double)in_w[in_beg + k] : NINF;
        odst[k] = o ? L.out_dst[out_beg + k] : 0;
        ow[k] = o ? (double)out_w[out_beg + k] : NINF;
        xi_r[k] = 0.0;
    }
    double flow_r = 0.0;                                  // hub -> this state
    // largest in / out degree of the graph: the arc loops below stop there (a
    // workgroup-uniform bound: phone-loop states have 2-3 of the BEER_SEG slots in use)
    const int kin = (int)block_max((double)(in_end - in_beg), hub);
    const int kout = (int)block_max((double)(out_end - out_beg), hub);

    // Members of the hub (phone-loop pivot: a few dozen sources / destinations) in
    // registers, one per lane: the lists and weights do not change over the frames,
    // and re-reading them from global memory put three dependent loads into every
    // step of the chain.  (`fits` false: more than kRegHubs hubs or more than 64
    // members -- the loops over the lists below.)
    int hm_src[kRegHubs], hm_dst[kRegHubs];
    T hw_src[kRegHubs], hw_dst[kRegHubs];          // (converted at use, as the loops do)
    bool fits = H <= kRegHubs;
#pragma unroll
    for (int h = 0; h < kRegHubs; ++h) {
        hm_src[h] = hm_dst[h] = -1;
        hw_src[h] = hw_dst[h] = ninf<T>();
        if (h < H) {
            const int sb = L.src_ptr[h], se = L.src_ptr[h + 1];
            const int db = L.dst_ptr[h], de = L.dst_ptr[h + 1];
            fits = fits && se - sb <= 64 && de - db <= 64;
            if (sb + lane < se) {
                hm_src[h] = L.src_list[sb + lane];
                hw_src[h] = hsw[hm_src[h]];
            }
            if (db + lane < de) {
                hm_dst[h] = L.dst_list[db + lane];
                hw_dst[h] = hdw[hm_dst[h]];
            }
        }
    }
    // log-sum-exp over the members of a hub held in registers: col[member] + weight
    auto hub_lse_reg = [&](const double* col, int member, T weight) {
        const double val = member >= 0 ? col[member] + (double)weight : NINF;
        const double m = wave_max(val);
        double r = m;
        if (m > NINF && m < PINF) r = m + flog<T>(wave_sum(fexp<T>(val - m)));
        return r;
    };

    // wave 0: log-sum-exp over the members of every hub
    // (vals[] are read from LDS `cur` or `lb` + per-member weight)
    auto hub_lse = [&](const double* col, const int32_t* ptr, const int32_t* list,
                       const T* w, const int (&hm)[kRegHubs], const T (&hw)[kRegHubs]) {
        if (fits) {
#pragma unroll
            for (int h = 0; h < kRegHubs; ++h)
                if (h < H) {
                    const double r = hub_lse_reg(col, hm[h], hw[h]);
                    if (lane == 0) hub[h] = r;
                }
            return;
        }
        for (int h = 0; h < H; ++h) {
            const int beg = ptr[h], end = ptr[h + 1];
            double m = NINF;
            for (int p = beg + lane; p < end; p += 64) {
                const double val = col[list[p]] + (double)w[list[p]];
                m = val > m ? val : m;
            }
            m = wave_max(m);
            double r = m;
            if (m > NINF && m < PINF) {
                double sm = 0.0;
                for (int p = beg + lane; p < end; p += 64)
                    sm += fexp<T>(col[list[p]] + (double)w[list[p]] - m);
                sm = wave_sum(sm);
                r = m + flog<T>(sm);
            }
            if (lane == 0) hub[h] = r;
        }
    };

    // ---- forward ----
    if (st) {
        const double a = (double)llh[j] + (double)((const T*)g.init)[j];
        cur[j] = a;
        alpha[j] = a;
    }
    __syncthreads();
    // The recursion is a chain of T dependent steps: everything a step reads from
    // global memory (its emission log-likelihood, in the backward pass also its
    // alpha) is loaded one step ahead, so that no memory round trip sits in the
    // chain (2 us per step before, most of it two exposed loads).
    double ll_next = st && T_ > 1 ? (double)llh[S + j] : 0.0;
    for (int64_t t = 1; t < T_; ++t) {
        const double ll = ll_next;
        if (st && t + 1 < T_) ll_next = (double)llh[(t + 1) * S + j];
        if (wave == 0 && H > 0) hub_lse(cur, L.src_ptr, L.src_list, hsw, hm_src, hw_src);
        double m = NINF, sm = 0.0;
        if (st) {
            // (absent arcs have weight -inf: their term is exp(-inf) = 0, no branch)
            double val[BEER_SEG];
#pragma unroll
            for (int k = 0; k < BEER_SEG; ++k)
                if (k < kin) {
                    val[k] = cur[isrc[k]] + iw[k];
                    m = val[k] > m ? val[k] : m;
                }
            if (m > NINF && m < PINF) {
#pragma unroll
                for (int k = 0; k < BEER_SEG; ++k)
                    if (k < kin) sm += fexp<T>(val[k] - m);
            }
        }
        __syncthreads();                                  // hub values visible; cur fully read
        if (st) {
            double lse = (m > NINF && m < PINF) ? m + flog<T>(sm) : m;
            if (hd >= 0) lse = logaddexp2<T>(lse, hub[hd] + hd_w);
            const double a = ll + lse;
            cur[j] = a;
            alpha[t * S + j] = a;
        }
        __syncthreads();
    }

    // ---- backward + posteriors ----
    if (st) cur[j] = (double)((const T*)g.final)[j];     // beta_{T-1}
    // alpha_t and llh_t of the frame in hand, loaded one iteration ahead; alpha_t
    // goes to LDS for the waves that reduce over all states
    double a_next = st ? alpha[(T_ - 1) * S + j] : NINF;
    double lt_next = st ? (double)llh[(T_ - 1) * S + j] : 0.0;
    __syncthreads();
    double ln_acc = 0.0;
    for (int64_t t = T_ - 1; t >= 0; --t) {
        const bool inner = t < T_ - 1;
        const double a_cur = a_next, lt_cur = lt_next;
        if (st) {
            al[j] = a_cur;
            if (t > 0) {
                a_next = alpha[(t - 1) * S + j];
                lt_next = (double)llh[(t - 1) * S + j];
            }
        }
        if (!inner) __syncthreads();                      // al visible (else: the barriers below)
        if (inner) {
            // beta_t(i) = lse(sparse: A_ij + lb_j ; hub: r_i + lse_s(w_s + lb_s))
            if (wave == 0 && H > 0) hub_lse(lb, L.dst_ptr, L.dst_list, hdw, hm_dst, hw_dst);
            double m = NINF, sm = 0.0;
            if (st) {
                double val[BEER_SEG];
#pragma unroll
                for (int k = 0; k < BEER_SEG; ++k)
                    if (k < kout) {
                        val[k] = ow[k] + lb[odst[k]];
                        m = val[k] > m ? val[k] : m;
                    }
                if (m > NINF && m < PINF) {
#pragma unroll
                    for (int k = 0; k < BEER_SEG; ++k)
                        if (k < kout) sm += fexp<T>(val[k] - m);
                }
            }
            __syncthreads();
            if (st) {
                double lse = (m > NINF && m < PINF) ? m + flog<T>(sm) : m;
                if (hs >= 0) lse = logaddexp2<T>(lse, hs_w + hub[hs]);
                cur[j] = lse;
            }
            __syncthreads();
        }
        // lognorm_t = lse_i(alpha_t(i) + beta_t(i)) by wave 0; forward hub values
        // H_h(t) = lse_e(alpha_t(e) + r_e) (for the hub flows) by wave 1
        const double ab = st ? a_cur + cur[j] : NINF;
        if (wave == 0) {
            double m = NINF;
            for (int p = lane; p < S; p += 64) {
                const double val = al[p] + cur[p];
                m = val > m ? val : m;
            }
            m = wave_max(m);
            double r = m;
            if (m > NINF && m < PINF) {
                double sm = 0.0;
                for (int p = lane; p < S; p += 64) sm += fexp<T>(al[p] + cur[p] - m);
                sm = wave_sum(sm);
                r = m + flog<T>(sm);
            }
            if (lane == 0) hub[kMaxHubs] = r;
        } else if (wave == 1 && inner && xi_sum && H > 0 && fits) {
#pragma unroll
            for (int h = 0; h < kRegHubs; ++h)
                if (h < H) {
                    const double r = hub_lse_reg(al, hm_src[h], hw_src[h]);
                    if (lane == 0) hub[kMaxHubs + 1 + h] = r;
                }
        } else if (wave == 1 && inner && xi_sum && H > 0) {
            for (int h = 0; h < H; ++h) {
                const int beg = L.src_ptr[h], end = L.src_ptr[h + 1];
                double m = NINF;
                for (int p = beg + lane; p < end; p += 64) {
                    const int e = L.src_list[p];
                    const double val = al[e] + (double)hsw[e];
                    m = val > m ? val : m;
                }
                m = wave_max(m);
                double r = m;
                if (m > NINF && m < PINF) {
                    double sm = 0.0;
                    for (int p = beg + lane; p < end; p += 64) {
                        const int e = L.src_list[p];
                        sm += fexp<T>(al[e] + (double)hsw[e] - m);
                    }
                    sm = wave_sum(sm);
                    r = m + flog<T>(sm);
                }
                if (lane == 0) hub[kMaxHubs + 1 + h] = r;
            }
        }
        __syncthreads();
        const double lognorm = hub[kMaxHubs];
        ln_acc += lognorm;
        if (st) {
            const double gv = fexp<T>(ab - lognorm);       // NaN if -inf - -inf, as the reference
            gam[t * S + j] = (T)gv;
            if (t == 0 && gamma0_sum) atomicAdd(gamma0_sum + j, gv);
            if (xi_sum && inner && lognorm > NINF) {
                // arcs t -> t+1 leaving this state, and the hub flow entering it
                const double ai = a_cur - lognorm;
#pragma unroll
                for (int k = 0; k < BEER_SEG; ++k) {
                    if (k < kout && ow[k] > NINF) {
                        const double val = fexp<T>(ai + ow[k] + lb[odst[k]]);
                        if (val == val) xi_r[k] += val;
                    }
                }
                if (hd >= 0) {
                    const double val = fexp<T>(hub[kMaxHubs + 1 + hd] + hd_w + lb[j] - lognorm);
                    if (val == val) flow_r += val;
                }
            }
        }
        __syncthreads();                                  // lb readers done
        if (st) lb[j] = lt_cur + cur[j];                  // for frame t-1
        __syncthreads();
    }
    if (lognorm_mean && tid == 0) lognorm_mean[u] = (T)(ln_acc / (double)T_);
    if (xi_sum && st) {
#pragma unroll
        for (int k = 0; k < BEER_SEG; ++k)
            if (ow[k] > NINF && xi_r[k] != 0.0) atomicAdd(xi_sum + (size_t)j * S + odst[k], xi_r[k]);
        if (hd >= 0 && hub_flow) atomicAdd(hub_flow + j, flow_r);
    }
}

// ---------------------------------------------------------------------------
// forward-backward, ONE WAVE per utterance (low-degree graphs with at most one
// hub, <= 256 states), in the SCALED LINEAR domain
// ---------------------------------------------------------------------------
// The workgroup kernels above are bound by instruction issue, not by the latency
// of their dependent steps, and so was the first one-wave kernel: a log-space step
// costs every state one exponential per arc, a logarithm, conversions around both
// and fp64 additions -- 344 vector instructions per frame at 120 states (rocprof,
// round 3), 77 % of the SIMDs' issue slots.  Here a wave owns an utterance alone
// -- lane l holds states l, l + 64, ... (SPL per lane) -- and the recursions run
// on probabilities, not on their logarithms:
//
//   a_t(j) = b_t(j) (sum_k W_in[j,k] a_{t-1}(src_k) + W_hd[j] H_{t-1}),
//                                              H_{t-1} = sum_{e in hub} W_hs[e] a_{t-1}(e)
//   beta'_t(i) = sum_k W_out[i,k] lb_{t+1}(dst_k) + W_hs[i] Hb_{t+1},   lb_t = b_t beta_t
//   gamma_t = a_t beta'_t / sum_j a_t(j) beta'_t(j)        (per frame, graph.py:304-307)
//
// with b_t(j) = exp(l_t(j) - m_t), m_t the float maximum of the frame's
// log-likelihoods over the states (any shift near it serves), W = exp(log-weight) in
// fp64 (absent arcs, absent hub links and lanes without a state: weight 0, so the
// unrolled DEG arc slots need no branch).  All of it is fp64 multiply-adds; ONE
// exponential per state, frame and direction is left.  After every step the column is
// multiplied by the power of two that brings its largest entry into [1/2, 1) -- exact,
// so nothing is rounded by the scaling and fp64's range (2^-1022) is what a state's
// mass relative to the frame's best may fall to before it is lost; posteriors are
// ratios inside one frame, the scales cancel.  log p(X) = sum_t m_t - log 2 sum_t
// shift_t + log sum_j a_{T-1}(j) final_j.
//  * exp(d) for float models: 2^n 2^f with n = rint(d log2 e) and f in [-1/2, 1/2]
//    formed in fp64, 2^f by v_exp_f32 -- relative error 1e-7 whatever |d| (rounding
//    d log2 e to float first would lose |d| 1e-7: states that are the only ones a
//    graph allows may lie hundreds below the frame's maximum);
//  * sums over the wave (hub, normaliser): float models add floats when the result
//    is comfortably inside float's range (the terms are <= 1 by the scaling), fp64
//    otherwise; fp64 models always fp64;
//  * the forward column is kept for the backward pass in fp64 (see A_t below);
//  * the trellis column lives in LDS without any barrier (the LDS operations of
//    one wave execute in order).
// FUSED: the pdf-id gather (modelset.py:140-146) with the acoustic scale
// (hmm.py:79) happens when the emission log-likelihoods are read, the scatter
// back to pdf ids (modelset.py:148-154, hmm.py:95) and the utterance's
// sum_t sum_s gamma * pc (hmm.py:87) when gamma is written: three launches and
// two round trips of the [frames, states] arrays less.
#ifndef BEER_FB_WAVES
#define BEER_FB_WAVES 4
#endif
constexpr int kWvWaves = BEER_FB_WAVES; // utterances (waves) per workgroup
#ifndef BEER_FB_PF
#define BEER_FB_PF 4
#endif
constexpr int kWvPF = BEER_FB_PF;      // steps of look-ahead of the global loads
#ifndef BEER_FB_RESCALE
#define BEER_FB_RESCALE 2
#endif
constexpr int kWvRescale = BEER_FB_RESCALE;   // a column is rescaled every kWvRescale-th frame (divides kWvPF)
static_assert(kWvPF % kWvRescale == 0, "the rescaled frames are fixed positions of the unrolled loop");
#ifndef BEER_FB_OCC
#define BEER_FB_OCC 4                  // waves per SIMD the linear-domain kernel is compiled for
#endif

typedef unsigned v2u_t __attribute__((ext_vector_type(2)));     // a 64-bit buffer word

// a wave-uniform 64-bit value, moved to SGPRs
template <typename V>
__device__ __forceinline__ V uniform64(V v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
    return __builtin_bit_cast(V, ((unsigned long long)hi << 32) | lo);
}
// Sum / maximum over the wave INTO AN SGPR: six DPP steps -- four inside the rows of 16 lanes,
// then row_bcast:15 and row_bcast:31 carry the rows' totals up to row 3 -- and one readlane
// of lane 63.  The generic all-reduce (common.h) spends a move, a lane swap and the
// operation on each of its two cross-row steps and leaves the result in every lane, where
// these kernels then had to fetch it for the scalar unit again: 12 vector instructions per
// reduction against 7, four reductions per frame.  Written as one asm block: hipcc does not
// fuse a row_bcast with row_mask into the operation (it emits v_mov_b32_dpp + the
// operation); the two wait states a DPP read needs after the VALU write of its source are
// the s_nops (the assembler does not add them inside inline asm).  All lanes take part.
#define BEER_DPP_CHAIN(OP)                                                                       \
    "s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
    "s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
    "s_nop 1\n\t" OP " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"     \
    "s_nop 1\n\t" OP " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"          \
    "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"                     \
    "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
__device__ __forceinline__ float wave_sum_scalar(float v) {
    asm volatile(BEER_DPP_CHAIN("v_add_f32_dpp") : "+v"(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ int wave_max_scalar(int v) {
    asm volatile(BEER_DPP_CHAIN("v_max_i32_dpp") : "+v"(v));
    return __builtin_amdgcn_readlane(v, 63);
}
#undef BEER_DPP_CHAIN
// maximum of a float over the wave (NaN if any lane holds one): on the order-preserving
// integer image of the floats, one v_max_i32 with a DPP operand per step instead of a move
// and two quieting operations around every v_max_f32
__device__ __forceinline__ float wave_fmax(float v) {
    int i = __builtin_bit_cast(int, v);
    i ^= (i >> 31) & 0x7fffffff;
    i = wave_max_scalar(i);
    i ^= (i >> 31) & 0x7fffffff;
    return __builtin_bit_cast(float, i);
}

template <typename T> struct Lin;
template <> struct Lin<float> {
    // exp(d) in fp64 range, relative error ~1e-7; d = -inf -> 0 (and NaN -> 0: the
    // kernel flags utterances with NaN log-likelihoods, see `gave_up`)
    static __device__ __forceinline__ double ex(double d) {
        const double y = d * 1.4426950408889634074;
        // (-inf and anything below 2^-1100 alike: 0 after the ldexp)
        const double yc = __builtin_fmax(y, -1100.0);
        const double n = __builtin_rint(yc);
        const double r = (double)__builtin_amdgcn_exp2f((float)(yc - n));
        return __builtin_amdgcn_ldexp(r, (int)n);
    }
    // sum over the wave of non-negative terms <= ~1
    static __device__ __forceinline__ double wsum(double v) {
        const float s = wave_sum_scalar((float)v);
        // (uniform; the float sum has lost nothing that matters unless it is tiny.  The test
        // is the scalar unit's: for a non-negative float, s > 1e-30f is an integer comparison
        // of the bit patterns; a NaN passes and is returned as it is)
        if (__builtin_bit_cast(int, s) > 0x0DA24260) return (double)s;
        return wave_sum(v);
    }
};
template <> struct Lin<double> {
    static __device__ __forceinline__ double ex(double d) { return exp(d); }
    static __device__ __forceinline__ double wsum(double v) { return wave_sum(v); }
};
// biased exponent of a non-negative double (0: zero or denormal)
__device__ __forceinline__ int expo_field(double v) {
    return (int)((__builtin_bit_cast(unsigned long long, v) >> 52) & 0x7ffull);
}

// OUTM (fused form): how a frame's posteriors go back to pdf ids -- 0: plain stores (every pdf id
// of the graph distinct); 1: global atomic adds into a zero-filled array (repeated ids: an alignment
// graph names a phone twice); 2: through a ROW IN LDS -- the wave adds its states' posteriors into
// an all-zero row of S_total entries (LDS atomics: repeated ids add up there), writes the whole
// row out with plain coalesced stores and clears it again.  No zero-filled output, no global
// atomics (1.07 M frames x 100 states of them cost the recipes' alignment-graph training 0.7 ms of
// a 6.4 ms epoch); needs S_total <= kWvRowMax.
constexpr int kWvRowMax = 512;
template <typename T, int SPL, int DEG, bool FUSED, bool XI, int OUTM>
__global__ __launch_bounds__(64 * kWvWaves, BEER_FB_OCC * 4 / kWvWaves) void fb_wave_kernel(
    beer_batch b, const T* __restrict__ pc, int S_total, T scale, double* __restrict__ alpha_ws,
    double* __restrict__ hubf_ws, T* __restrict__ out, T resp_scale,
    double* __restrict__ xi_sum, double* __restrict__ gamma0_sum, double* __restrict__ hub_flow,
    double* __restrict__ utt_llh, T* __restrict__ lognorm_mean, T* __restrict__ frame_llh) {
    typedef Lin<T> R;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slot = blockIdx.x * kWvWaves + wave;
    if (slot >= b.nutt) return;                           // (no workgroup barrier below)
    const int u = b.order ? __builtin_amdgcn_readfirstlane(b.order[slot]) : slot;
    const int gid = b.graph_id[u];
    const beer_graph g = b.graphs[gid];
    const beer_graph_lowdeg L = *g.lowdeg;
    const int S = g.n_states;
    const bool has_hub = L.n_hubs > 0;
    // (wave-uniform by construction: in SGPRs, so that the frame loops and the row pointers
    // are the scalar unit's work)
    const int64_t f0 = uniform64(b.frame_off[u]);
    const int T_ = __builtin_amdgcn_readfirstlane((int)(b.frame_off[u + 1] - f0));
    if (T_ <= 0) return;
    constexpr int NS = 64 * SPL;                          // LDS slots per column
    // byte addresses in LDS: cur[NS] (a_{t-1}), lb[NS] (b_{t+1} beta_{t+1})
    const int cur0 = wave * (2 * NS * 8), lb0 = cur0 + NS * 8;
    auto lds = [&](int addr) -> double& { return *reinterpret_cast<double*>(smem + addr); };
    // (OUTM == 2) the wave's output row, behind the columns of all waves
    T* orow_lds = reinterpret_cast<T*>(smem + kWvWaves * (2 * NS * 8)) + wave * kWvRowMax;
    if constexpr (OUTM == 2) {
        for (int e = lane; e < S_total; e += 64) orow_lds[e] = (T)0;
    }
    const T* in_w = (const T*)L.in_w;
    const T* out_w = (const T*)L.out_w;
    const T* hsw = (const T*)L.hub_src_w;
    const T* hdw = (const T*)L.hub_dst_w;
    const T* init = (const T*)g.init;
    const T* fin = (const T*)g.final;

    // ---- the lane's states (lanes without one: weights 0, their own slots) ----
    int own[SPL], isrc[SPL][DEG], odst[SPL][DEG];          // LDS byte addresses
    double iw[SPL][DEG], ow[SPL][DEG], hs_w[SPL], hd_w[SPL], fin_w[SPL];
    bool st[SPL];
    int64_t ll_off[SPL], a_off[SPL];                       // element offsets per frame 0
    double xi_r[XI ? SPL : 1][DEG], flow_r[SPL];
    const int32_t* ids = FUSED ? b.pdf_ids + b.pdf_off[gid] : nullptr;
#pragma unroll
    for (int p = 0; p < SPL; ++p) {
        const int j = lane + 64 * p;
        st[p] = j < S;
        own[p] = cur0 + 8 * j;
        hs_w[p] = hd_w[p] = fin_w[p] = 0.0;
        flow_r[p] = 0.0;
        int ib = 0, ie = 0, ob = 0, oe = 0, id = 0;
        if (st[p]) {
            ib = L.in_ptr[j]; ie = L.in_ptr[j + 1];
            ob = L.out_ptr[j]; oe = L.out_ptr[j + 1];
            if (has_hub && L.hub_src_id[j] >= 0) hs_w[p] = exp((double)hsw[j]);
            if (has_hub && L.hub_dst_id[j] >= 0) hd_w[p] = exp((double)hdw[j]);
            fin_w[p] = exp((double)fin[j]);
            if (FUSED) id = ids[j];
        }
        // (lanes without a state: the pdf of the graph's OWN state 0 -- always a valid element, and
        //  never another graph's column, whose NaN is not this utterance's)
        ll_off[p] = st[p] ? (FUSED ? (int64_t)id : (int64_t)j) : (FUSED ? (int64_t)ids[0] : 0);
        a_off[p] = j;
#pragma unroll
        for (int k = 0; k < DEG; ++k) {
            const bool a = ib + k < ie, o = ob + k < oe;
            isrc[p][k] = a ? cur0 + 8 * L.in_src[ib + k] : own[p];
            iw[p][k] = a ? exp((double)in_w[ib + k]) : 0.0;
            odst[p][k] = o ? lb0 + 8 * L.out_dst[ob + k] : own[p] + NS * 8;
            ow[p][k] = o ? exp((double)out_w[ob + k]) : 0.0;
            if (XI) xi_r[p][k] = 0.0;
        }
    }
    // the hub's members, one per lane
    int hm_src = own[0], hm_dst = own[0] + NS * 8;
    double hw_src = 0.0, hw_dst = 0.0;
    if (has_hub) {
        const int sb = L.src_ptr[0], se = L.src_ptr[1], db = L.dst_ptr[0], de = L.dst_ptr[1];
        if (sb + lane < se) {
            const int e = L.src_list[sb + lane];
            hm_src = cur0 + 8 * e;
            hw_src = exp((double)hsw[e]);
        }
        if (db + lane < de) {
            const int e = L.dst_list[db + lane];
            hm_dst = lb0 + 8 * e;
            hw_dst = exp((double)hdw[e]);
        }
    }
    // The wave's LDS writes are visible to its later reads in program order; the
    // compiler must keep that order (no instruction is emitted for this).
#define BEER_WAVE_ORDER() do { __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); } while (0)

    const T* llh = FUSED ? pc + f0 * (int64_t)S_total : pc + b.llh_off[u];
    const int64_t ll_stride = FUSED ? S_total : S;
    // Every global access of the recursions goes through a buffer descriptor of ONE ROW
    // (frame) of its array -- a uniform base the scalar unit advances per frame, a 32-bit
    // lane offset that never changes, and the hardware's range check instead of a branch:
    // lanes without a state carry an offset past the row, their loads return 0 and their
    // stores are dropped (the loop had a saveexec / branch pair around every store and two
    // 64-bit address computations per access).
    constexpr int kOob = 0x7fffffff;
    auto row_of = [&](const void* base, int64_t elems, int elem_bytes) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0,
                                                 (int)(elems * elem_bytes), 0x00020000);
    };
    int vo_ll[SPL], vo_a[SPL], vo_out[SPL];
#pragma unroll
    for (int p = 0; p < SPL; ++p) {
        vo_ll[p] = (int)(ll_off[p] * (int64_t)sizeof(T));          // (lanes without a state: element 0)
        vo_a[p] = st[p] ? (int)(a_off[p] * 8) : kOob;
        vo_out[p] = st[p] ? (int)((FUSED ? ll_off[p] : a_off[p]) * (int64_t)sizeof(T)) : kOob;
    }
    const int vo_lane0 = lane == 0 ? 0 : kOob;
    // (kept in fp64 also for float models: a float column loses a state below 2^-126 of
    // the frame's best -- 87 nats, which evidence after the frame does make up for on real
    // utterances: the real-dimension parity test failed with float columns.  Measured: 6 %
    // less time for the 480 bytes per frame, but every utterance that then trips the
    // normaliser test pays the log-space kernel on top)
    typedef double A_t;
    A_t* alpha = reinterpret_cast<A_t*>(alpha_ws + b.llh_off[u]);   // scaled forward columns
    auto al_row = [&](int64_t t) { return row_of(alpha + t * S, S, 8); };
    constexpr double kMinNorm = 0x1p-800;
    double* hubf = hubf_ws + f0;                          // forward hub value per frame
    const __amdgpu_buffer_rsrc_t hub_rs = row_of(hubf, T_, 8);
    auto load_word = [&](__amdgpu_buffer_rsrc_t r, int vo) -> T {
        if constexpr (sizeof(T) == 4)
            return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b32(r, vo, 0, 0));
        else
            return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b64(r, vo, 0, 0));
    };
    auto store_word = [&](T v, __amdgpu_buffer_rsrc_t r, int vo) {
        if constexpr (sizeof(T) == 4)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, vo, 0, 0);
        else
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u_t, v), r, vo, 0, 0);
    };
    auto load_f64 = [&](__amdgpu_buffer_rsrc_t r, int vo) -> double {
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, vo, 0, 0));
    };
    auto store_f64 = [&](double v, __amdgpu_buffer_rsrc_t r, int vo) {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u_t, v), r, vo, 0, 0);
    };
    auto ll_row = [&](int64_t t) { return row_of(llh + t * ll_stride, ll_stride, sizeof(T)); };
    // (the utterance's slice of the per-frame values; no buffer: every store out of range)
    const __amdgpu_buffer_rsrc_t frame_rs =
        row_of(frame_llh ? frame_llh + f0 : nullptr, frame_llh ? T_ : 0, sizeof(T));
    auto load_ll = [&](int64_t t, int p) -> T {
        const T v = load_word(ll_row(t), vo_ll[p]);
        return FUSED ? scale * v : v;
    };
    // the shift of a block of NF frames: the float maximum over their states (a frame
    // whose own maximum lies below it starts out smaller by that factor, which the
    // scaling behind it takes out again).  A NaN log-likelihood flags the utterance: the
    // log-space kernel then propagates it as the reference does.
    bool gave_up = false;
    auto frames_max = [&](const T (*ll)[SPL], int nf) -> double {
        float m = -INFINITY;
        bool bad = false;
        for (int k = 0; k < nf; ++k)
#pragma unroll
            for (int p = 0; p < SPL; ++p) {
                m = __builtin_fmaxf(m, (float)ll[k][p]);  // (lanes without a state read state 0's)
                // (only the graph's own pdf columns count, as in the reference: the value a
                //  state-less lane read belongs to another utterance's graph)
                bad |= st[p] && ll[k][p] != ll[k][p];
            }
        m = wave_fmax(m);
        gave_up |= __builtin_amdgcn_ballot_w64(bad) != 0;
        return uniform64(m > -INFINITY && m < INFINITY ? (double)m : 0.0);
    };
    // Giving up.  fp64 holds a state's mass down to 2^-1022 of the column's largest; what
    // falls below is lost, and harmless unless the OTHER pass weights exactly those
    // states up by as much (evidence before and after a frame contradicting each other by
    // hundreds of nats).  Every column whose largest entry, before its scaling, is at
    // least 2^-800, and every frame whose normaliser sum_j a_t(j) beta'_t(j) is at least
    // kMinNorm = 2^-800, has lost nothing that matters (the lost terms are 2^-222 of what
    // is kept); an
    // utterance that fails either test is flagged (the slot behind its hub values) and
    // redone in log space by fb_wave_log_kernel, and this kernel adds nothing of it to
    // the accumulators.
    // column * 2^shift with its largest entry in [1/2, 1); returns shift (0: all zero)
    auto rescale = [&](double (&v)[SPL]) -> int {
        int e = 0;
#pragma unroll
        for (int p = 0; p < SPL; ++p) { const int ep = expo_field(v[p]); e = ep > e ? ep : e; }
        e = wave_max_scalar(e);
        gave_up |= e < 1023 - 800 || e >= 0x7ff;
        const int sh = (e > 0 && e < 0x7ff) ? 1022 - e : 0;
#pragma unroll
        for (int p = 0; p < SPL; ++p) v[p] = __builtin_amdgcn_ldexp(v[p], sh);
        return sh;
    };

    // ---- forward ----
    // The loads of a step are issued PF steps ahead (a ring of registers, the loops
    // unrolled PF times so that every ring index is a constant): with the recursion on
    // multiply-adds a step is shorter than a trip to HBM, and one step of look-ahead left
    // every wave waiting for its 480 bytes.
    constexpr int PF = kWvPF;
    double m_sum = 0.0, sh_sum = 0.0;                     // log scale of the stored columns
    T ll_ring[PF][SPL];
    {
        // (the first shift is taken from log-likelihood + initial log-probability)
        double a[SPL], l0[SPL];
        float m = -INFINITY;
        bool bad0 = false;
#pragma unroll
        for (int p = 0; p < SPL; ++p) {
            l0[p] = st[p] ? (double)load_ll(0, p) + (double)init[lane + 64 * p] : neg_inf();
            m = __builtin_fmaxf(m, (float)l0[p]);
            bad0 |= l0[p] != l0[p];
        }
        gave_up |= __builtin_amdgcn_ballot_w64(bad0) != 0;
        m = wave_fmax(m);
        const double m0 = m > -INFINITY && m < INFINITY ? (double)m : 0.0;
#pragma unroll
        for (int p = 0; p < SPL; ++p) a[p] = st[p] ? R::ex(l0[p] - m0) : 0.0;
        const int sh = rescale(a);
        m_sum += m0;
        sh_sum += (double)sh;
#pragma unroll
        for (int p = 0; p < SPL; ++p) {
            lds(own[p]) = a[p];
            store_f64(a[p], al_row(0), vo_a[p]);
        }
#pragma unroll
        for (int k = 0; k < PF; ++k)
#pragma unroll
            for (int p = 0; p < SPL; ++p) ll_ring[k][p] = load_ll(1 + k < T_ ? 1 + k : T_ - 1, p);
    }
    BEER_WAVE_ORDER();
    auto forward_step = [&](int64_t t, const T (&ll)[SPL], double mt, bool scale_now) {
        double hub = 0.0;
        if (has_hub) {
            hub = R::wsum(lds(hm_src) * hw_src);
            store_f64(hub, hub_rs, vo_lane0 == 0 ? (int)(8 * (t - 1)) : kOob);   // H(t-1): flows t-1 -> t
        }
        double a[SPL];
#pragma unroll
        for (int p = 0; p < SPL; ++p) {
            double pred = hub * hd_w[p];
#pragma unroll
            for (int k = 0; k < DEG; ++k) pred = __builtin_fma(lds(isrc[p][k]), iw[p][k], pred);
            a[p] = R::ex((double)ll[p] - mt) * pred;       // (lanes without a state: weights 0)
        }
        // (the column is brought back to [1/2, 1) on every kWvRescale-th frame: between two
        // rescalings it can only shrink, by what the frames' likelihoods differ from their
        // block's maximum -- a column that runs out of fp64's range on the way is all zero
        // at the next rescaling and flags the utterance)
        if (scale_now) {
            const int sh = rescale(a);
            if constexpr (!FUSED) sh_sum += (double)sh;        // (the fused launch reports no log p(X))
        }
        if constexpr (!FUSED) m_sum += mt;
        BEER_WAVE_ORDER();                                // every read of the column is done
        {
            const __amdgpu_buffer_rsrc_t ar = al_row(t);
#pragma unroll
            for (int p = 0; p < SPL; ++p) {
                lds(own[p]) = a[p];
                store_f64(a[p], ar, vo_a[p]);
            }
        }
        BEER_WAVE_ORDER();
    };
    for (int t0 = 1; t0 < T_; t0 += PF) {
        const double mt = frames_max(ll_ring, PF);        // (frames past the end: the last one again)
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int t = t0 + k;
            if (t >= T_) break;
            T ll[SPL];
#pragma unroll
            for (int p = 0; p < SPL; ++p) {
                ll[p] = ll_ring[k][p];
                ll_ring[k][p] = load_ll(t + PF < T_ ? t + PF : T_ - 1, p);
            }
            forward_step(t, ll, mt, (k % kWvRescale) == kWvRescale - 1);
        }
    }

    // ---- backward + posteriors ----
    // frame T-1 first (beta = final, nothing to recurse), then the loop
    double llh_acc = 0.0, log_px = 0.0;
    double lb_own[SPL], g0v[SPL];                  // (g0v: the posteriors of the frame done last, 0)
    A_t a_ring[PF][SPL];                           // frames t - 1 - k
    T lt_ring[PF][SPL];
    double hf_ring[PF];
    // beta'_t (before its scaling) -> posteriors of frame t, flows of the arcs t -> t+1,
    // then lb_t = b_t beta_t for frame t - 1
    auto finish_frame = [&](int64_t t, const double (&a_cur)[SPL], const T (&lt_cur)[SPL],
                            double (&beta)[SPL], const double (&lbd)[SPL][DEG], double hf_cur,
                            bool inner, double mt, bool scale_now) {
        double gq[SPL], gsum = 0.0;
#pragma unroll
        for (int p = 0; p < SPL; ++p) {
            gq[p] = a_cur[p] * beta[p];                    // (lanes without a state: 0)
            gsum += gq[p];
        }
        const double norm = R::wsum(gsum);                 // per frame (graph.py:304-307)
        if (t == T_ - 1) log_px = norm;
        const bool ok = norm >= kMinNorm && norm < __builtin_huge_val();
        gave_up |= !ok;
        // (a frame that fails the test flags the utterance: whatever it adds to the
        // accumulators below is never flushed, so nothing here needs to look at `ok`)
        // 1 / norm: the hardware's estimate and two Newton steps (a flagged frame's value
        // is not used)
        double inv = __builtin_amdgcn_rcp(norm);
        inv = __builtin_fma(__builtin_fma(-norm, inv, 1.0), inv, inv);
        inv = __builtin_fma(__builtin_fma(-norm, inv, 1.0), inv, inv);
        const __amdgpu_buffer_rsrc_t orow = FUSED
            ? row_of(out + (f0 + t) * (int64_t)S_total, S_total, sizeof(T))
            : row_of(out + b.llh_off[u] + t * S, S, sizeof(T));
        const double hfi = hf_cur * inv;
        T fval = 0;                                        // this lane's share of sum_s gamma l
#pragma unroll
        for (int p = 0; p < SPL; ++p) {
            const double gv = gq[p] * inv;
            g0v[p] = gv;
            if (FUSED) {
                const T gT = (T)gv;
                if constexpr (OUTM == 1) {
                    // (repeated pdf ids: added, not stored)
                    if (st[p]) atomicAdd(out + (f0 + t) * (int64_t)S_total + ll_off[p], resp_scale * gT);
                } else if constexpr (OUTM == 2) {
                    if (st[p])
                        __hip_atomic_fetch_add(orow_lds + ll_off[p], resp_scale * gT, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WAVEFRONT);
                } else {
                    store_word(resp_scale * gT, orow, vo_out[p]);
                }
                // (lanes without a state read state 0's log-likelihood: not theirs to add)
                const T prod = st[p] ? lt_cur[p] * gT : (T)0;
                llh_acc += (double)prod;
                fval += prod;
            } else {
                store_word((T)gv, orow, vo_out[p]);
            }
            if (inner) {
                if (XI) {
                    const double ai = a_cur[p] * inv;
#pragma unroll
                    for (int k = 0; k < DEG; ++k) xi_r[p][k] = __builtin_fma(ai * ow[p][k], lbd[p][k], xi_r[p][k]);
                }
                if (has_hub && hub_flow)
                    flow_r[p] = __builtin_fma(hfi * hd_w[p], lb_own[p], flow_r[p]);
            }
        }
        if constexpr (FUSED && OUTM == 2) {
            // the row is complete: out it goes (all S_total entries, zeros included), and clear
            BEER_WAVE_ORDER();
            for (int e = lane; e < S_total; e += 64) {
                const T v = orow_lds[e];
                orow_lds[e] = (T)0;
                store_word(v, orow, e * (int)sizeof(T));
            }
            BEER_WAVE_ORDER();
        }
        if (FUSED && frame_llh) {
            // the frame's expected log-likelihood sum_s gamma_ts l_ts (hmm.py:87) while both
            // factors are in registers: one wave reduction instead of a pass over two [T, S] arrays
            T fsum;
            if constexpr (sizeof(T) == 4) fsum = wave_sum_scalar(fval);
            else fsum = wave_sum(fval);
            store_word(fsum, frame_rs, lane == 0 ? (int)(t * (int64_t)sizeof(T)) : kOob);
        }
        if (t > 0) {
            if (scale_now) (void)rescale(beta);
#pragma unroll
            for (int p = 0; p < SPL; ++p)
                lb_own[p] = R::ex((double)lt_cur[p] - mt) * beta[p];     // for frame t - 1
        }
        BEER_WAVE_ORDER();                                 // lb fully read
#pragma unroll
        for (int p = 0; p < SPL; ++p) lds(own[p] + NS * 8) = lb_own[p];
        BEER_WAVE_ORDER();
    };
    auto load_back = [&](int64_t tq, A_t (&av)[SPL], T (&lv)[SPL], double& hv) {
        const int64_t tc = tq > 0 ? tq : 0;                // (clamped: loaded, never used)
        const __amdgpu_buffer_rsrc_t ar = al_row(tc);
#pragma unroll
        for (int p = 0; p < SPL; ++p) {
            av[p] = load_f64(ar, vo_a[p]);                  // (lanes without a state: 0)
            lv[p] = load_ll(tc, p);
        }
        hv = has_hub ? load_f64(hub_rs, (int)(8 * tc)) : 0.0;
    };
    {
        const int64_t t = T_ - 1;
        double a_cur[SPL], beta[SPL], lbd[SPL][DEG];
        T lt_cur[SPL];
#pragma unroll
        for (int p = 0; p < SPL; ++p) {
            a_cur[p] = load_f64(al_row(t), vo_a[p]);
            lt_cur[p] = load_ll(t, p);
            beta[p] = fin_w[p];
            lb_own[p] = 0.0;
#pragma unroll
            for (int k = 0; k < DEG; ++k) lbd[p][k] = 0.0;
        }
#pragma unroll
        for (int k = 0; k < PF; ++k) load_back(t - 1 - k, a_ring[k], lt_ring[k], hf_ring[k]);
        T one[1][SPL];
#pragma unroll
        for (int p = 0; p < SPL; ++p) one[0][p] = lt_cur[p];
        finish_frame(t, a_cur, lt_cur, beta, lbd, 0.0, false, frames_max(one, 1), true);
    }
    auto backward_step = [&](int64_t t, const A_t (&a_t)[SPL], const T (&lt_cur)[SPL], double hf_cur,
                             double mt, bool scale_now) {
        double a_cur[SPL], beta[SPL], lbd[SPL][DEG];
        double hub = 0.0;
        if (has_hub) hub = R::wsum(lds(hm_dst) * hw_dst);
#pragma unroll
        for (int p = 0; p < SPL; ++p) {
            a_cur[p] = (double)a_t[p];
            double acc = hub * hs_w[p];
#pragma unroll
            for (int k = 0; k < DEG; ++k) {
                lbd[p][k] = lds(odst[p][k]);
                acc = __builtin_fma(lbd[p][k], ow[p][k], acc);
            }
            beta[p] = acc;
        }
        finish_frame(t, a_cur, lt_cur, beta, lbd, hf_cur, true, mt, scale_now);
    };
    for (int t0 = T_ - 2; t0 >= 0; t0 -= PF) {
        const double mt = frames_max(lt_ring, PF);        // (frames before the start: frame 0 again)
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int t = t0 - k;
            if (t < 0) break;
            A_t a_t[SPL];
            T lt_cur[SPL];
#pragma unroll
            for (int p = 0; p < SPL; ++p) { a_t[p] = a_ring[k][p]; lt_cur[p] = lt_ring[k][p]; }
            const double hf_cur = hf_ring[k];              // H(t)
            load_back(t - PF, a_ring[k], lt_ring[k], hf_ring[k]);
            backward_step(t, a_t, lt_cur, hf_cur, mt, (k % kWvRescale) == kWvRescale - 1);
        }
    }
#undef BEER_WAVE_ORDER
    if (lane == 0) hubf[T_ - 1] = gave_up ? 1.0 : 0.0;     // (hub values: slots 0 .. T-2)
    if (gave_up) return;
    if (gamma0_sum) {
        // the posteriors of frame 0 (the frame the backward pass finished with)
#pragma unroll
        for (int p = 0; p < SPL; ++p)
            if (st[p]) atomicAdd(gamma0_sum + a_off[p], g0v[p]);
    }
    // log p(X): every frame's log-normaliser (graph.py:304-326 averages T equal numbers)
    if (lognorm_mean && lane == 0)
        lognorm_mean[u] = (T)(m_sum - sh_sum * 0.69314718055994530942 + log(log_px));
    if (FUSED && utt_llh) {
        llh_acc = wave_sum(llh_acc);
        if (lane == 0) atomicAdd(utt_llh + u, llh_acc);
    }
#pragma unroll
    for (int p = 0; p < SPL; ++p) {
        if (!st[p]) continue;
        const int j = lane + 64 * p;
        if (XI && xi_sum) {
#pragma unroll
            for (int k = 0; k < DEG; ++k)
                if (ow[p][k] > 0.0 && xi_r[p][k] != 0.0)
                    atomicAdd(xi_sum + (size_t)j * S + (odst[p][k] - lb0) / 8, xi_r[p][k]);
        }
        if (hub_flow && hd_w[p] > 0.0) atomicAdd(hub_flow + j, flow_r[p]);
    }
}

// ---------------------------------------------------------------------------
// ... and in LOG SPACE, for the utterances the linear-domain kernel gave up on (its
// flag in hub_ws): round 3's kernel.  A log-space step costs every state one
// exponential per arc, a logarithm, conversions around both and fp64 additions -- 344
// vector instructions per frame at 120 states against ~215, most of them multiply-adds
// -- but its range is unlimited, like the reference's (graph.py:270-326):
//  * every log-sum-exp is "relative to an approximate maximum": the terms stay
//    fp64, their maximum is taken in float (any m near the maximum gives the same
//    sum), differences to it are formed in fp64 and exponentiated, summed and log'ed
//    in the model's precision;
//  * absent arcs, absent hub links and lanes without a state carry the weight -inf;
//  * the forward values are kept in fp64 (their size grows with the utterance).
// ---------------------------------------------------------------------------
// arithmetic of a log-sum-exp relative to an approximate maximum
template <typename T> struct Rel;
template <> struct Rel<float> {
    typedef float r_t;
    static __device__ __forceinline__ float down(double v) { return (float)v; }
    static __device__ __forceinline__ float mx(float a, float b) { return __builtin_fmaxf(a, b); }
    static __device__ __forceinline__ float wmax(float v) {
        return wave_detail::allreduce(v, [](float x, float y) { return __builtin_fmaxf(x, y); });
    }
    static __device__ __forceinline__ float ex(double d) {       // exp(d), d <~ 0
        return __builtin_amdgcn_exp2f((float)d * 1.44269504088896340736f);
    }
    static __device__ __forceinline__ double lg(float s) {        // log(s)
        return (double)(__builtin_amdgcn_logf(s) * 0.69314718055994530942f);
    }
};
template <> struct Rel<double> {
    typedef double r_t;
    static __device__ __forceinline__ double down(double v) { return v; }
    static __device__ __forceinline__ double mx(double a, double b) { return b > a ? b : a; }
    static __device__ __forceinline__ double wmax(double v) { return wave_max(v); }
    static __device__ __forceinline__ double ex(double d) { return exp(d); }
    static __device__ __forceinline__ double lg(double s) { return log(s); }
};
// the maximum as an fp64 offset: 0 when every term is -inf (the sum is then 0 and
// its log -inf, as it must be)
template <typename R>
__device__ __forceinline__ double rel_base(R m) {
    return m > (R)-INFINITY ? (double)m : 0.0;
}

template <typename T, int SPL, int DEG, bool FUSED, bool XI>
__global__ __launch_bounds__(64 * kWvWaves) void fb_wave_log_kernel(
    beer_batch b, const T* __restrict__ pc, int S_total, T scale, double* __restrict__ alpha_ws,
    double* __restrict__ hubf_ws, T* __restrict__ out, T resp_scale, int atomic_out,
    double* __restrict__ xi_sum, double* __restrict__ gamma0_sum, double* __restrict__ hub_flow,
    double* __restrict__ utt_llh, T* __restrict__ lognorm_mean, T* __restrict__ frame_llh) {
    typedef Rel<T> R;
    typedef typename R::r_t r_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slot = blockIdx.x * kWvWaves + wave;
    if (slot >= b.nutt) return;                           // (no workgroup barrier below)
    const int u = b.order ? __builtin_amdgcn_readfirstlane(b.order[slot]) : slot;
    const int gid = b.graph_id[u];
    const beer_graph g = b.graphs[gid];
    const beer_graph_lowdeg L = *g.lowdeg;
    const int S = g.n_states;
    const bool has_hub = L.n_hubs > 0;
    const int64_t f0 = b.frame_off[u], T_ = b.frame_off[u + 1] - f0;
    if (T_ <= 0) return;
    if (hubf_ws[f0 + T_ - 1] == 0.0) return;             // the linear-domain kernel did this one
    const double NINF = neg_inf();
    constexpr int NS = 64 * SPL;                          // LDS slots per column
    // byte addresses in LDS: cur[NS] (alpha_{t-1}), lb[NS] (llh_{t+1} + beta_{t+1})
    const int cur0 = wave * (2 * NS * 8), lb0 = cur0 + NS * 8;
    auto lds = [&](int addr) -> double& { return *reinterpret_cast<double*>(smem + addr); };
    // (atomic_out == 2) the wave's output row, behind the columns of all waves: see fb_wave_kernel
    const bool rows_out = FUSED && atomic_out == 2;
    T* orow_lds = reinterpret_cast<T*>(smem + kWvWaves * (2 * NS * 8)) + wave * kWvRowMax;
    if (rows_out)
        for (int e = lane; e < S_total; e += 64) orow_lds[e] = (T)0;
    auto flush_row = [&](int64_t t) {
        __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory");
        T* dst = out + (f0 + t) * (int64_t)S_total;
        for (int e = lane; e < S_total; e += 64) {
            const T v = orow_lds[e];
            orow_lds[e] = (T)0;
            dst[e] = v;
        }
        __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory");
    };
    const T* in_w = (const T*)L.in_w;
    const T* out_w = (const T*)L.out_w;
    const T* hsw = (const T*)L.hub_src_w;
    const T* hdw = (const T*)L.hub_dst_w;
    const T* init = (const T*)g.init;
    const T* fin = (const T*)g.final;

    // ---- the lane's states (lanes without one: weights -inf, their own slots) ----
    int own[SPL], isrc[SPL][DEG], odst[SPL][DEG];          // LDS byte addresses
    T iw[SPL][DEG], ow[SPL][DEG], hs_w[SPL], hd_w[SPL], fin_w[SPL];
    bool st[SPL];
    int64_t ll_off[SPL], a_off[SPL];                       // element offsets per frame 0
    double xi_r[XI ? SPL : 1][DEG], flow_r[SPL];
    const int32_t* ids = FUSED ? b.pdf_ids + b.pdf_off[gid] : nullptr;
#pragma unroll
    for (int p = 0; p < SPL; ++p) {
        const int j = lane + 64 * p;
        st[p] = j < S;
        own[p] = cur0 + 8 * j;
        hs_w[p] = hd_w[p] = fin_w[p] = ninf<T>();
        flow_r[p] = 0.0;
        int ib = 0, ie = 0, ob = 0, oe = 0, id = 0;
        if (st[p]) {
            ib = L.in_ptr[j]; ie = L.in_ptr[j + 1];
            ob = L.out_ptr[j]; oe = L.out_ptr[j + 1];
            if (has_hub && L.hub_src_id[j] >= 0) hs_w[p] = hsw[j];
            if (has_hub && L.hub_dst_id[j] >= 0) hd_w[p] = hdw[j];
            fin_w[p] = fin[j];
            if (FUSED) id = ids[j];
        }
        // (lanes without a state: the pdf of the graph's OWN state 0 -- always a valid element, and
        //  never another graph's column, whose NaN is not this utterance's)
        ll_off[p] = st[p] ? (FUSED ? (int64_t)id : (int64_t)j) : (FUSED ? (int64_t)ids[0] : 0);
        a_off[p] = j;
#pragma unroll
        for (int k = 0; k < DEG; ++k) {
            const bool a = ib + k < ie, o = ob + k < oe;
            isrc[p][k] = a ? cur0 + 8 * L.in_src[ib + k] : own[p];
            iw[p][k] = a ? in_w[ib + k] : ninf<T>();
            odst[p][k] = o ? lb0 + 8 * L.out_dst[ob + k] : own[p] + NS * 8;
            ow[p][k] = o ? out_w[ob + k] : ninf<T>();
            if (XI) xi_r[p][k] = 0.0;
        }
    }
    // the hub's members, one per lane
    int hm_src = own[0], hm_dst = own[0] + NS * 8;
    T hw_src = ninf<T>(), hw_dst = ninf<T>();
    if (has_hub) {
        const int sb = L.src_ptr[0], se = L.src_ptr[1], db = L.dst_ptr[0], de = L.dst_ptr[1];
        if (sb + lane < se) {
            const int e = L.src_list[sb + lane];
            hm_src = cur0 + 8 * e;
            hw_src = hsw[e];
        }
        if (db + lane < de) {
            const int e = L.dst_list[db + lane];
            hm_dst = lb0 + 8 * e;
            hw_dst = hdw[e];
        }
    }
    // log-sum-exp over the wave of one value per lane
    auto wave_lse = [&](double val) {
        const double m = rel_base(R::wmax(R::down(val)));
        return m + R::lg(wave_sum(R::ex(val - m)));
    };
    // log-sum-exp of the lane's DEG arc terms and its hub term
    auto lane_lse = [&](const double (&v)[DEG], double vh) {
        r_t m = R::down(vh);
#pragma unroll
        for (int k = 0; k < DEG; ++k) m = R::mx(m, R::down(v[k]));
        const double mb = rel_base(m);
        r_t sm = R::ex(vh - mb);
#pragma unroll
        for (int k = 0; k < DEG; ++k) sm += R::ex(v[k] - mb);
        return mb + R::lg(sm);
    };
    // The wave's LDS writes are visible to its later reads in program order; the
    // compiler must keep that order (no instruction is emitted for this).
#define BEER_WAVE_ORDER() do { __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); } while (0)

    const T* llh = FUSED ? pc + f0 * (int64_t)S_total : pc + b.llh_off[u];
    const int64_t ll_stride = FUSED ? S_total : S;
    double* alpha = alpha_ws + b.llh_off[u];
    double* hubf = hubf_ws + f0;                          // forward hub value per frame
    auto load_ll = [&](int64_t t, int p) -> T {
        const T v = llh[t * ll_stride + ll_off[p]];       // (lanes without a state: element 0)
        return FUSED ? scale * v : v;
    };

    if (FUSED && atomic_out == 1) {
        // (the fast kernel may have added posteriors of some of the utterance's frames
        // before it gave up: the rows are this utterance's alone; whole rows -- atomic_out == 2 --
        // are simply written again)
        for (int64_t e = lane; e < T_ * (int64_t)S_total; e += 64) out[f0 * (int64_t)S_total + e] = (T)0;
        __threadfence();
    }
    // ---- forward ----
    T ll_next[SPL];
    bool bad = false;                                     // a NaN log-likelihood (below)
#pragma unroll
    for (int p = 0; p < SPL; ++p) {
        const double a = st[p] ? (double)load_ll(0, p) + (double)init[lane + 64 * p] : NINF;
        bad |= a != a;
        lds(own[p]) = a;
        if (st[p]) alpha[a_off[p]] = a;
        ll_next[p] = load_ll(T_ > 1 ? 1 : 0, p);
    }
    BEER_WAVE_ORDER();
    for (int64_t t = 1; t < T_; ++t) {
        T ll[SPL];
#pragma unroll
        for (int p = 0; p < SPL; ++p) {
            ll[p] = ll_next[p];
            bad |= st[p] && ll[p] != ll[p];               // (the graph's own pdf columns only)
            ll_next[p] = load_ll(t + 1 < T_ ? t + 1 : t, p);
        }
        double hub = NINF;
        if (has_hub) {
            hub = wave_lse(lds(hm_src) + (double)hw_src);
            if (lane == 0) hubf[t - 1] = hub;             // H(t-1): flows of the arcs t-1 -> t
        }
        double a[SPL];
#pragma unroll
        for (int p = 0; p < SPL; ++p) {
            double v[DEG];
#pragma unroll
            for (int k = 0; k < DEG; ++k) v[k] = lds(isrc[p][k]) + (double)iw[p][k];
            a[p] = (double)ll[p] + lane_lse(v, hub + (double)hd_w[p]);
        }
        BEER_WAVE_ORDER();                                // every read of the column is done
#pragma unroll
        for (int p = 0; p < SPL; ++p) {
            lds(own[p]) = st[p] ? a[p] : NINF;
            if (st[p]) alpha[t * S + a_off[p]] = a[p];
        }
        BEER_WAVE_ORDER();
    }

    if (__builtin_amdgcn_ballot_w64(bad) != 0) {
        // A NaN log-likelihood anywhere in the utterance.  The reference adds it to EVERY
        // entry of its dense transition matrix (NaN + -inf = NaN: graph.py:274-277,
        // 283-286), so every forward value after the frame and every backward value before
        // it is NaN: all state posteriors of the utterance are NaN, its transition
        // posteriors 0 (NaN -> 0, graph.py:319-321), its log-normaliser NaN.  The sparse
        // recursion would let the NaN travel along arcs only; say what the reference says.
        const T nan_t = (T)__builtin_nan("");
        for (int64_t t = 0; t < T_; ++t) {
#pragma unroll
            for (int p = 0; p < SPL; ++p) {
                if (!st[p]) continue;
                if (FUSED) {
                    T* dst = out + (f0 + t) * (int64_t)S_total + ll_off[p];
                    if (rows_out) orow_lds[ll_off[p]] = nan_t;
                    else if (atomic_out) atomicAdd(dst, nan_t);
                    else *dst = nan_t;
                } else {
                    out[b.llh_off[u] + t * S + a_off[p]] = nan_t;
                }
            }
            if (rows_out) flush_row(t);
        }
#pragma unroll
        for (int p = 0; p < SPL; ++p)
            if (st[p] && gamma0_sum) atomicAdd(gamma0_sum + a_off[p], __builtin_nan(""));
        if (FUSED && frame_llh)
            for (int64_t t = lane; t < T_; t += 64) frame_llh[f0 + t] = nan_t;
        if (lane == 0) {
            if (lognorm_mean) lognorm_mean[u] = nan_t;
            if (FUSED && utt_llh) atomicAdd(utt_llh + u, __builtin_nan(""));
        }
        return;
    }

    // ---- backward + posteriors ----
    // frame T-1 first (beta = final, nothing to recurse), then the loop
    double ln_acc = 0.0, llh_acc = 0.0;
    double lb_own[SPL], a_next[SPL];
    T lt_next[SPL];
    double hf_next = NINF;
    auto finish_frame = [&](int64_t t, const double (&a_cur)[SPL], const T (&lt_cur)[SPL],
                            const double (&beta)[SPL], const double (&vout)[SPL][DEG],
                            double hf_cur, bool inner) {
        double ab[SPL];
        r_t m = (r_t)-INFINITY;
#pragma unroll
        for (int p = 0; p < SPL; ++p) {
            ab[p] = a_cur[p] + beta[p];                    // (lanes without a state: -inf)
            m = R::mx(m, R::down(ab[p]));
        }
        const double mb = rel_base(R::wmax(m));
        r_t sm = 0;
#pragma unroll
        for (int p = 0; p < SPL; ++p) sm += R::ex(ab[p] - mb);
        const double lognorm = mb + R::lg(wave_sum(sm));   // per frame (graph.py:304-307)
        ln_acc += lognorm;
        const bool ok = lognorm > NINF;
        T fval = 0;
#pragma unroll
        for (int p = 0; p < SPL; ++p) {
            // NaN when alpha + beta and lognorm are both -inf, as in the reference
            const double gv = st[p] ? (double)R::ex(ab[p] - lognorm) : 0.0;
            if (st[p]) {
                if (FUSED) {
                    const T gT = (T)gv;
                    T* dst = out + (f0 + t) * (int64_t)S_total + ll_off[p];
                    if (rows_out)
                        __hip_atomic_fetch_add(orow_lds + ll_off[p], resp_scale * gT, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WAVEFRONT);
                    else if (atomic_out) atomicAdd(dst, resp_scale * gT);
                    else *dst = resp_scale * gT;
                    llh_acc += (double)(lt_cur[p] * gT);
                    fval += lt_cur[p] * gT;
                } else {
                    out[b.llh_off[u] + t * S + a_off[p]] = (T)gv;
                }
                if (t == 0 && gamma0_sum) atomicAdd(gamma0_sum + a_off[p], gv);
            }
            if (inner) {
                if (XI) {
                    const double ai = a_cur[p] - lognorm;
#pragma unroll
                    for (int k = 0; k < DEG; ++k) {
                        const double val = (double)R::ex(ai + vout[p][k]);
                        xi_r[p][k] += (ok && val == val) ? val : 0.0;
                    }
                }
                if (has_hub && hub_flow) {
                    const double val = (double)R::ex(hf_cur + (double)hd_w[p] + lb_own[p] - lognorm);
                    flow_r[p] += (ok && val == val) ? val : 0.0;
                }
            }
        }
        if (rows_out) flush_row(t);
        if (FUSED && frame_llh) {
            const T fsum = wave_sum(fval);
            if (lane == 0) frame_llh[f0 + t] = fsum;
        }
        BEER_WAVE_ORDER();                                 // lb fully read
#pragma unroll
        for (int p = 0; p < SPL; ++p) {
            lb_own[p] = st[p] ? (double)lt_cur[p] + beta[p] : NINF;     // for frame t - 1
            lds(own[p] + NS * 8) = lb_own[p];
        }
        BEER_WAVE_ORDER();
    };
    {
        const int64_t t = T_ - 1;
        double a_cur[SPL], beta[SPL], vout[SPL][DEG];
        T lt_cur[SPL];
#pragma unroll
        for (int p = 0; p < SPL; ++p) {
            a_cur[p] = st[p] ? alpha[t * S + a_off[p]] : NINF;
            lt_cur[p] = load_ll(t, p);
            beta[p] = (double)fin_w[p];
            lb_own[p] = NINF;
#pragma unroll
            for (int k = 0; k < DEG; ++k) vout[p][k] = NINF;
            a_next[p] = (st[p] && t > 0) ? alpha[(t - 1) * S + a_off[p]] : NINF;
            lt_next[p] = load_ll(t > 0 ? t - 1 : 0, p);
        }
        if (has_hub && t > 0) hf_next = hubf[t - 1];
        finish_frame(t, a_cur, lt_cur, beta, vout, NINF, false);
    }
    for (int64_t t = T_ - 2; t >= 0; --t) {
        double a_cur[SPL], beta[SPL], vout[SPL][DEG];
        T lt_cur[SPL];
        const double hf_cur = hf_next;                    // H(t)
#pragma unroll
        for (int p = 0; p < SPL; ++p) {
            a_cur[p] = a_next[p];
            lt_cur[p] = lt_next[p];
            const int64_t tp = t > 0 ? t - 1 : 0;
            a_next[p] = st[p] ? alpha[tp * S + a_off[p]] : NINF;
            lt_next[p] = load_ll(tp, p);
        }
        if (has_hub) hf_next = hubf[t > 0 ? t - 1 : 0];
        double hub = NINF;
        if (has_hub) hub = wave_lse(lds(hm_dst) + (double)hw_dst);
#pragma unroll
        for (int p = 0; p < SPL; ++p) {
#pragma unroll
            for (int k = 0; k < DEG; ++k) vout[p][k] = (double)ow[p][k] + lds(odst[p][k]);
            beta[p] = lane_lse(vout[p], (double)hs_w[p] + hub);
        }
        finish_frame(t, a_cur, lt_cur, beta, vout, hf_cur, true);
    }
#undef BEER_WAVE_ORDER
    if (lognorm_mean && lane == 0) lognorm_mean[u] = (T)(ln_acc / (double)T_);
    if (FUSED && utt_llh) {
        llh_acc = wave_sum(llh_acc);
        if (lane == 0) atomicAdd(utt_llh + u, llh_acc);
    }
#pragma unroll
    for (int p = 0; p < SPL; ++p) {
        if (!st[p]) continue;
        const int j = lane + 64 * p;
        if (XI && xi_sum) {
#pragma unroll
            for (int k = 0; k < DEG; ++k)
                if (ow[p][k] > ninf<T>() && xi_r[p][k] != 0.0)
                    atomicAdd(xi_sum + (size_t)j * S + (odst[p][k] - lb0) / 8, xi_r[p][k]);
        }
        if (hub_flow && hd_w[p] > ninf<T>()) atomicAdd(hub_flow + j, flow_r[p]);
    }
}

// ---------------------------------------------------------------------------
// Viterbi
// ---------------------------------------------------------------------------
// The recursion is a chain of T dependent steps per utterance, so nothing a step
// needs may come from global memory inside the chain: the graph's arc lists live
// in LDS, the emission log-likelihoods of step t + 1 are loaded during step t, and
// the back-trace walks the back-pointers through LDS, kViterbiChunk frames at a
// time (one coalesced load per chunk instead of one dependent load per frame).
constexpr int kViterbiChunk = 32;

constexpr int kViterbiThreads = 512;

template <typename T>
__global__ __launch_bounds__(kViterbiThreads) void viterbi_kernel(
    beer_batch b, const T* __restrict__ pc_llhs, int32_t* __restrict__ bt_ws,
    int64_t* __restrict__ path, int map_pdf, int arcs_in_lds) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int u = blockIdx.x, tid = threadIdx.x, nt_ = blockDim.x;
    const int gid = b.graph_id[u];
    const beer_graph g = b.graphs[gid];
    const int S = g.n_states, A = g.n_arcs;
    const int64_t T_ = b.frame_off[u + 1] - b.frame_off[u];
    if (T_ <= 0) return;
    const T* llh = pc_llhs + b.llh_off[u];
    int32_t* bt = bt_ws + b.llh_off[u];
    int64_t* out = path + b.frame_off[u];
    const T* init = (const T*)g.init;
    const T* fin = (const T*)g.final;
    T* cur = reinterpret_cast<T*>(smem);
    T* nxt = cur + b.max_states;
    int32_t* btc = reinterpret_cast<int32_t*>(nxt + b.max_states);     // [chunk][S]
    int32_t* l_ptr = btc + (size_t)kViterbiChunk * b.max_states;       // [S + 1]
    int32_t* l_src = l_ptr + b.max_states + 1;                         // [A]
    T* l_w = reinterpret_cast<T*>(l_src + b.max_arcs + ((b.max_states + 1 + b.max_arcs) & 1));
    const int32_t* in_ptr = g.in_ptr;
    const int32_t* in_src = g.in_src;
    const T* in_w = (const T*)g.in_w;
    if (arcs_in_lds) {
        for (int j = tid; j <= S; j += nt_) l_ptr[j] = g.in_ptr[j];
        for (int e = tid; e < A; e += nt_) { l_src[e] = g.in_src[e]; l_w[e] = in_w[e]; }
        in_ptr = l_ptr; in_src = l_src; in_w = l_w;
    }
    for (int j = tid; j < S; j += nt_) cur[j] = llh[j] + init[j];
    __syncthreads();
    // (one state per thread in the common case: the prefetch register is per thread)
    const bool one = S <= nt_;
    // Four lanes per state when the workgroup has them: the in-arcs of a state are
    // dealt to the lanes of a quad and the four (best, source) pairs combined by DPP
    // -- larger value, on ties the smaller source index: the first maximum in source
    // order, as one lane scanning the whole list finds it.  A phone-loop state that
    // follows the pivot has an arc from every phone (40 here) while most have 2: the
    // longest list sets the time of the step.
    const bool quad = 4 * S <= nt_;
    const int qj = tid >> 2, qp = tid & 3;
    T ll_next = (T)0;
    if (quad) { if (qj < S && T_ > 1) ll_next = llh[S + qj]; }
    else if (one && tid < S && T_ > 1) ll_next = llh[S + tid];
    for (int64_t t = 1; t < T_; ++t) {
        const T ll_cur = ll_next;
        if (quad) {
            if (qj < S && t + 1 < T_) ll_next = llh[(t + 1) * S + qj];
            T best = ninf<T>();
            int arg = 0;
            if (qj < S) {
                for (int e = in_ptr[qj] + qp; e < in_ptr[qj + 1]; e += 4) {
                    const int src = in_src[e];
                    const T v = cur[src] + in_w[e];
                    if (v > best) { best = v; arg = src; }
                }
            }
            {
                const T ob = wave_detail::dpp<T, 0xB1>(best);
                const int oa = (int)wave_detail::dpp_word<0xB1>((unsigned)arg);
                if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
            }
            {
                const T ob = wave_detail::dpp<T, 0x4E>(best);
                const int oa = (int)wave_detail::dpp_word<0x4E>((unsigned)arg);
                if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
            }
            if (qj < S && qp == 0) {
                bt[t * S + qj] = arg;
                nxt[qj] = ll_cur + best;
            }
            __syncthreads();
            T* tmp = cur; cur = nxt; nxt = tmp;
            continue;
        }
        if (one && tid < S && t + 1 < T_) ll_next = llh[(t + 1) * S + tid];
        for (int j = tid; j < S; j += nt_) {
            // argmax_i(omega_i + A_ij): first index wins; all -inf -> 0
            // (torch.argmax over the dense column, graph.py:337-338).
            T best = ninf<T>();
            int arg = 0;
            for (int e = in_ptr[j]; e < in_ptr[j + 1]; ++e) {
                const int src = in_src[e];
                const T v = cur[src] + in_w[e];
                if (v > best) { best = v; arg = src; }
            }
            bt[t * S + j] = arg;
            // hypothesis[j, arg]: when no arc is finite the dense entry is
            // omega_0 + A_0j = -inf (or NaN-free -inf + finite)
            nxt[j] = (one ? ll_cur : llh[t * S + j]) + best;
        }
        __syncthreads();
        T* tmp = cur; cur = nxt; nxt = tmp;
    }
    // final state (thread 0), then the back-trace chunk by chunk
    int* s_state = reinterpret_cast<int*>(nxt);             // nxt is free now
    if (tid == 0) {
        T best = ninf<T>();
        int arg = 0;
        for (int j = 0; j < S; ++j) {
            const T v = cur[j] + fin[j];
            if (v > best) { best = v; arg = j; }
        }
        s_state[0] = arg;
    }
    const int32_t* ids = b.pdf_ids + b.pdf_off[gid];
    for (int64_t hi = T_ - 1; hi >= 0; hi -= kViterbiChunk) {
        const int64_t lo = hi - kViterbiChunk + 1 > 0 ? hi - kViterbiChunk + 1 : 0;   // frames lo .. hi
        __syncthreads();                                    // bt stores / previous chunk done
        // back-pointers of frames max(lo, 1) .. hi (frame 0 has none)
        const int64_t first = lo > 0 ? lo : 1;
        for (int64_t e = tid; e < (hi - first + 1) * S; e += nt_) btc[e] = bt[first * S + e];
        __syncthreads();
        if (tid == 0) {
            int st = s_state[0];
            for (int64_t t = hi; t >= lo; --t) {
                out[t] = map_pdf ? (int64_t)ids[st] : (int64_t)st;
                if (t > 0) st = btc[(t - first) * S + st];
            }
            s_state[0] = st;
        }
    }
}

template <typename T>
__global__ void path_post_kernel(beer_batch b, const int64_t* __restrict__ path,
                                 T* __restrict__ gamma, double* __restrict__ xi_sum,
                                 double* __restrict__ gamma0_sum) {
    const int u = blockIdx.x;
    const int S = b.graphs[b.graph_id[u]].n_states;
    const int64_t f0 = b.frame_off[u], nt = b.frame_off[u + 1] - f0;
    T* g = gamma + b.llh_off[u];
    const int64_t* p = path + f0;
    for (int64_t idx = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; idx < nt * S;
         idx += (int64_t)gridDim.y * blockDim.x)
        g[idx] = (T)((idx % S) == p[idx / S] ? 1 : 0);
    if (blockIdx.y == 0) {
        if (xi_sum)
            for (int64_t t = threadIdx.x; t < nt - 1; t += blockDim.x)
                atomicAdd(xi_sum + (size_t)p[t] * S + p[t + 1], 1.0);
        if (gamma0_sum && threadIdx.x == 0 && nt > 0) atomicAdd(gamma0_sum + p[0], 1.0);
    }
}

// ---------------------------------------------------------------------------
// Per-frame transition posteriors in the reference's layout, [T-1, S, S]
// (graph.py:308-323), for callers that ask for that tensor (small inputs: it is
// 8 S^2 bytes per frame).  From what a forward-backward call leaves behind --
// alpha (fp64 workspace), gamma -- and the dense transition matrix:
//   xi_t(i,j)  ~  alpha_t(i) + A_ij + llh_{t+1}(j) + beta_{t+1}(j),
//   beta_{t+1}(j) = ln gamma_{t+1}(j) - alpha_{t+1}(j) + const,
// normalised per frame; a frame whose terms are all -inf gives 0 (the
// reference's NaN -> 0).  One workgroup per frame.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void xi_dense_kernel(int64_t nframes, int S,
                                                       const double* __restrict__ alpha,
                                                       const T* __restrict__ llh,
                                                       const T* __restrict__ gamma,
                                                       const T* __restrict__ trans,
                                                       T* __restrict__ out) {
    __shared__ double red[8];
    const int64_t t = blockIdx.x;
    if (t + 1 >= nframes) return;
    const double NINF = neg_inf();
    const double* a0 = alpha + t * S;
    const double* a1 = alpha + (t + 1) * S;
    const T* l1 = llh + (t + 1) * S;
    const T* g1 = gamma + (t + 1) * S;
    T* o = out + t * (int64_t)S * S;
    auto term = [&](int e) {
        const int i = e / S, j = e - i * S;
        const double gj = (double)g1[j];
        if (!(gj > 0.0)) return NINF;
        return a0[i] + (double)trans[e] + (double)l1[j] + log(gj) - a1[j];
    };
    double m = NINF;
    for (int e = threadIdx.x; e < S * S; e += blockDim.x) {
        const double v = term(e);
        m = v > m ? v : m;
    }
    m = block_max(m, red);
    const bool finite = m > NINF && m < __builtin_huge_val();
    double sm = 0.0;
    if (finite)
        for (int e = threadIdx.x; e < S * S; e += blockDim.x) sm += exp(term(e) - m);
    sm = block_sum(sm, red);
    for (int e = threadIdx.x; e < S * S; e += blockDim.x)
        o[e] = finite && sm > 0.0 ? (T)(exp(term(e) - m) / sm) : (T)0;
}

// ---- launchers -------------------------------------------------------------

template <typename T>
int gather_launch(const beer_batch* b, int S_total, const void* pc_all, double scale, void* out,
                  void* stream) {
    BEER_REQUIRE(b && b->nutt >= 0 && S_total >= 1);
    if (b->nutt == 0) return BEER_OK;
    hipLaunchKernelGGL(gather_kernel<T>, dim3(b->nutt, 4), dim3(256), 0, as_stream(stream), *b,
                       S_total, (const T*)pc_all, (T)scale, (T*)out);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

template <typename T>
int scatter_launch(const beer_batch* b, int S_total, const void* pc, const void* gamma,
                   double scale, void* sr, void* exp_llh, double* utt_llh, void* stream) {
    BEER_REQUIRE(b && b->nutt >= 0 && S_total >= 1);
    if (b->nutt == 0) return BEER_OK;
    if (!exp_llh) {
        hipLaunchKernelGGL(scatter_flat_kernel<T>, dim3(b->nutt, 8), dim3(256), 0,
                           as_stream(stream), *b, S_total, (const T*)pc, (const T*)gamma,
                           (T)scale, (T*)sr, utt_llh);
        BEER_LAUNCH_CHECK();
        return BEER_OK;
    }
    hipLaunchKernelGGL(scatter_kernel<T>, dim3(b->nutt, 8), dim3(256), 0, as_stream(stream), *b,
                       S_total, (const T*)pc, (const T*)gamma, (T)scale, (T*)sr, (T*)exp_llh,
                       utt_llh);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

// The log-space flag of an utterance is the slot behind its hub values (hub_ws[first
// frame + T - 1]; the recursions write slots 0 .. T-2 only).  BEER_OPT_FB_LOG sets it
// for every utterance instead of running the linear-domain kernel;
// beer_hmm_fb_log_count reads the flags back.
__global__ void fb_flag_all_kernel(beer_batch b, double* __restrict__ hubf_ws) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= b.nutt) return;
    const int64_t f1 = b.frame_off[u + 1];
    if (f1 > b.frame_off[u]) hubf_ws[f1 - 1] = 1.0;
}
__global__ void fb_flag_count_kernel(beer_batch b, const double* __restrict__ hubf_ws,
                                     int64_t* __restrict__ count) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    bool flagged = false;
    if (u < b.nutt) {
        const int64_t f1 = b.frame_off[u + 1];
        flagged = f1 > b.frame_off[u] && hubf_ws[f1 - 1] != 0.0;
    }
    const unsigned long long m = __builtin_amdgcn_ballot_w64(flagged);
    if ((threadIdx.x & 63) == 0 && m)
        atomicAdd(reinterpret_cast<unsigned long long*>(count), (unsigned long long)__builtin_popcountll(m));
}

// graphs the one-wave-per-utterance kernel takes: low-degree images with at most one
// hub of at most 64 members a side, <= 4 states per lane (the host layer fills the
// batch's max_degree / max_hubs / max_hub_members; 0 = unknown: not taken)
constexpr int kWvMaxStates = 256;
inline bool wave_fb_ok(const beer_batch* b) {
    return b->all_lowdeg && b->max_states <= kWvMaxStates && b->max_degree >= 1 &&
           b->max_degree <= BEER_SEG && b->max_hubs <= 1 && b->max_hub_members <= 64;
}

template <typename T, bool FUSED>
int wave_fb_launch(const beer_batch* b, const T* pc, int S_total, T scale, double* alpha_ws,
                   double* hub_ws, T* out, T resp_scale, int atomic_out, double* xi_sum,
                   double* gamma0_sum, double* hub_flow, double* utt_llh, T* lognorm_mean,
                   T* frame_llh, hipStream_t s) {
    const dim3 grid((unsigned)((b->nutt + kWvWaves - 1) / kWvWaves)), block(64 * kWvWaves);
    const bool xi = !FUSED && xi_sum != nullptr;
    const int spl = b->max_states <= 64 ? 1 : (b->max_states <= 128 ? 2 : 4);
    const int deg = b->max_degree <= 2 ? 2 : (b->max_degree <= 4 ? 4 : 8);
    const bool rows = FUSED && atomic_out == 2;
    if (rows && S_total > kWvRowMax) return BEER_EINVAL;
    const size_t lds = (size_t)kWvWaves * 2 * 64 * spl * sizeof(double) +
                       (rows ? (size_t)kWvWaves * kWvRowMax * sizeof(T) : 0);
    const bool all_log = beer::option(BEER_OPT_FB_LOG) != 0;
    // the linear-domain kernel, then the log-space one for the utterances it flagged
#define BEER_WV(SPL_, DEG_, XI_)                                                                \
    do {                                                                                        \
        if (all_log)                                                                            \
            hipLaunchKernelGGL(fb_flag_all_kernel, dim3((unsigned)((b->nutt + 255) / 256)),     \
                               dim3(256), 0, s, *b, hub_ws);                                    \
        else if (rows)                                                                          \
            hipLaunchKernelGGL((fb_wave_kernel<T, SPL_, DEG_, FUSED, XI_, FUSED ? 2 : 0>), grid, \
                               block, lds, s, *b, pc, S_total, scale, alpha_ws, hub_ws, out,    \
                               resp_scale, xi_sum, gamma0_sum, hub_flow, utt_llh, lognorm_mean, \
                               frame_llh);                                                     \
        else if (FUSED && atomic_out)                                                           \
            hipLaunchKernelGGL((fb_wave_kernel<T, SPL_, DEG_, FUSED, XI_, FUSED ? 1 : 0>), grid, \
                               block, lds, s, *b, pc, S_total, scale, alpha_ws, hub_ws, out,    \
                               resp_scale, xi_sum, gamma0_sum, hub_flow, utt_llh, lognorm_mean, \
                               frame_llh);                                                     \
        else                                                                                    \
            hipLaunchKernelGGL((fb_wave_kernel<T, SPL_, DEG_, FUSED, XI_, 0>), grid, block,     \
                               lds, s, *b, pc, S_total, scale, alpha_ws, hub_ws, out,           \
                               resp_scale, xi_sum, gamma0_sum, hub_flow, utt_llh, lognorm_mean, \
                               frame_llh);                                                     \
        hipLaunchKernelGGL((fb_wave_log_kernel<T, SPL_, DEG_, FUSED, XI_>), grid, block, lds,  \
                           s, *b, pc, S_total, scale, alpha_ws, hub_ws, out, resp_scale,       \
                           atomic_out, xi_sum, gamma0_sum, hub_flow, utt_llh, lognorm_mean,    \
                           frame_llh);                                                         \
    } while (0)
#define BEER_WV_DEG(SPL_, XI_)                                                                  \
    do {                                                                                        \
        if (deg == 2) BEER_WV(SPL_, 2, XI_);                                                    \
        else if (deg == 4) BEER_WV(SPL_, 4, XI_);                                               \
        else BEER_WV(SPL_, 8, XI_);                                                             \
    } while (0)
#define BEER_WV_SPL(XI_)                                                                        \
    do {                                                                                        \
        if (spl == 1) BEER_WV_DEG(1, XI_);                                                      \
        else if (spl == 2) BEER_WV_DEG(2, XI_);                                                 \
        else BEER_WV_DEG(4, XI_);                                                               \
    } while (0)
    if constexpr (FUSED) {
        BEER_WV_SPL(false);
    } else {
        if (xi) BEER_WV_SPL(true);
        else BEER_WV_SPL(false);
    }
#undef BEER_WV_SPL
#undef BEER_WV_DEG
#undef BEER_WV
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

}  // namespace

extern "C" {

int beer_hmm_posteriors_fused(int dtype, const beer_batch* b, int S_total, const void* pc_all,
                              double scale, double* alpha_ws, double* hub_ws, void* state_resps,
                              int atomic_out, double* gamma0_sum, double* hub_flow,
                              double* utt_llh, void* frame_llh, void* stream) {
    BEER_REQUIRE(b && b->nutt >= 0 && b->max_states >= 1 && S_total >= 1);
    BEER_REQUIRE(dtype == BEER_F32 || dtype == BEER_F64);
    BEER_REQUIRE(wave_fb_ok(b));
    if (b->nutt == 0) return BEER_OK;
    BEER_REQUIRE(pc_all && alpha_ws && hub_ws && state_resps);
    hipStream_t s = as_stream(stream);
    if (dtype == BEER_F32)
        return wave_fb_launch<float, true>(b, (const float*)pc_all, S_total, (float)scale,
                                           alpha_ws, hub_ws, (float*)state_resps, (float)scale,
                                           atomic_out, nullptr, gamma0_sum, hub_flow, utt_llh,
                                           nullptr, (float*)frame_llh, s);
    return wave_fb_launch<double, true>(b, (const double*)pc_all, S_total, scale, alpha_ws,
                                        hub_ws, (double*)state_resps, scale, atomic_out, nullptr,
                                        gamma0_sum, hub_flow, utt_llh, nullptr, (double*)frame_llh, s);
}

int beer_hmm_gather(int dtype, const beer_batch* batch_h, int S_total, const void* pc_all,
                    double scale, void* pc_llhs, void* stream) {
    BEER_DISPATCH(dtype, gather_launch, batch_h, S_total, pc_all, scale, pc_llhs, stream);
}

int beer_hmm_scatter(int dtype, const beer_batch* batch_h, int S_total, const void* pc_llhs,
                     const void* gamma, double scale, void* state_resps, void* exp_llh,
                     double* utt_llh, void* stream) {
    BEER_DISPATCH(dtype, scatter_launch, batch_h, S_total, pc_llhs, gamma, scale, state_resps,
                  exp_llh, utt_llh, stream);
}

int beer_hmm_forward_backward(int dtype, const beer_batch* b, const void* pc_llhs,
                              double* alpha_ws, double* hub_ws, void* gamma, double* xi_sum,
                              double* gamma0_sum, double* hub_flow, void* lognorm_mean,
                              void* stream) {
    BEER_REQUIRE(b && b->nutt >= 0 && b->max_states >= 1 && b->max_states <= 32767);
    BEER_REQUIRE(dtype == BEER_F32 || dtype == BEER_F64);
    if (b->nutt == 0) return BEER_OK;
    hipStream_t s = as_stream(stream);
    if (wave_fb_ok(b) && hub_ws && (!xi_sum || hub_flow)) {
        // one wave per utterance, no barriers
        if (dtype == BEER_F32)
            return wave_fb_launch<float, false>(b, (const float*)pc_llhs, b->max_states, 1.f,
                                                alpha_ws, hub_ws, (float*)gamma, 1.f, 0, xi_sum,
                                                gamma0_sum, hub_flow, nullptr,
                                                (float*)lognorm_mean, nullptr, s);
        return wave_fb_launch<double, false>(b, (const double*)pc_llhs, b->max_states, 1.0,
                                             alpha_ws, hub_ws, (double*)gamma, 1.0, 0, xi_sum,
                                             gamma0_sum, hub_flow, nullptr,
                                             (double*)lognorm_mean, nullptr, s);
    }
    if (b->all_lowdeg && b->max_states <= kLdThreads && (!xi_sum || hub_flow)) {
        // factorised low-degree recursion: one thread per state
        const size_t lds = ((size_t)3 * b->max_states + 4 * kMaxHubs + 8) * sizeof(double);
        // at least two waves (wave 1 recomputes the forward hub values)
        const int threads = b->max_states <= 128 ? 128 : (b->max_states <= 256 ? 256 : 512);
        if (dtype == BEER_F32)
            hipLaunchKernelGGL(fb_lowdeg_kernel<float>, dim3(b->nutt), dim3(threads), lds, s,
                               *b, (const float*)pc_llhs, alpha_ws, (float*)gamma, xi_sum,
                               gamma0_sum, hub_flow, (float*)lognorm_mean);
        else
            hipLaunchKernelGGL(fb_lowdeg_kernel<double>, dim3(b->nutt), dim3(threads), lds, s,
                               *b, (const double*)pc_llhs, alpha_ws, (double*)gamma, xi_sum,
                               gamma0_sum, hub_flow, (double*)lognorm_mean);
        BEER_LAUNCH_CHECK();
        return BEER_OK;
    }
    const size_t lds = fb_lds_bytes(b, dtype, xi_sum != nullptr, false);
    const bool big = lds > kLdsBytes;
#define BEER_FB(T_, BIG_, LDS_, GRID_)                                                           \
    do {                                                                                         \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fb_kernel<T_, BIG_>),            \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, beer::kMaxDynLds);      \
        hipLaunchKernelGGL((fb_kernel<T_, BIG_>), dim3(GRID_), dim3(kFbThreads), LDS_, s, *b,    \
                           (const T_*)pc_llhs, alpha_ws, (T_*)gamma, xi_sum, gamma0_sum,         \
                           (T_*)lognorm_mean, hub_ws);                                           \
    } while (0)
    if (!big) {
        if (dtype == BEER_F32) BEER_FB(float, false, lds, b->nutt);
        else BEER_FB(double, false, lds, b->nutt);
    } else {
        // arc lists beyond a CU's LDS: per-arc scratch in `hub_ws`
        // (beer_hmm_fb_scratch_doubles), the per-state arrays must still fit
        const size_t lds_big = fb_lds_bytes(b, dtype, xi_sum != nullptr, true);
        BEER_REQUIRE(lds_big <= kLdsBytes && hub_ws);
        const int grid = b->nutt < kFbBigBlocks ? b->nutt : kFbBigBlocks;
        if (dtype == BEER_F32) BEER_FB(float, true, lds_big, grid);
        else BEER_FB(double, true, lds_big, grid);
    }
#undef BEER_FB
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

int beer_hmm_fb_log_count(const beer_batch* b, const double* hub_ws, int64_t* count,
                          void* stream) {
    BEER_REQUIRE(b && b->nutt >= 0 && count);
    BEER_REQUIRE(wave_fb_ok(b));
    if (b->nutt == 0) return BEER_OK;
    BEER_REQUIRE(hub_ws);
    hipLaunchKernelGGL(fb_flag_count_kernel, dim3((unsigned)((b->nutt + 255) / 256)), dim3(256),
                       0, as_stream(stream), *b, hub_ws, count);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

size_t beer_hmm_fb_scratch_doubles(int dtype, const beer_batch* b, int want_xi) {
    if (!b || (dtype != BEER_F32 && dtype != BEER_F64) || b->nutt <= 0) return 0;
    if (fb_lds_bytes(b, dtype, want_xi != 0, false) <= kLdsBytes) return 0;
    const size_t blocks = b->nutt < kFbBigBlocks ? b->nutt : kFbBigBlocks;
    return blocks * (2 * (size_t)b->max_arcs + (size_t)b->max_segs);
}

int beer_hmm_trans_posteriors(int dtype, int64_t T, int S, const double* alpha, const void* llhs,
                              const void* gamma, const void* trans, void* xi, void* stream) {
    BEER_REQUIRE(T >= 0 && S >= 1 && (dtype == BEER_F32 || dtype == BEER_F64));
    if (T <= 1) return BEER_OK;
    BEER_REQUIRE(alpha && llhs && gamma && trans && xi);
    if (dtype == BEER_F32)
        hipLaunchKernelGGL(xi_dense_kernel<float>, dim3((unsigned)(T - 1)), dim3(256), 0,
                           as_stream(stream), T, S, alpha, (const float*)llhs, (const float*)gamma,
                           (const float*)trans, (float*)xi);
    else
        hipLaunchKernelGGL(xi_dense_kernel<double>, dim3((unsigned)(T - 1)), dim3(256), 0,
                           as_stream(stream), T, S, alpha, (const double*)llhs,
                           (const double*)gamma, (const double*)trans, (double*)xi);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

int beer_hmm_viterbi(int dtype, const beer_batch* b, const void* pc_llhs, int32_t* bt_ws,
                     int64_t* path, int map_pdf, void* stream) {
    BEER_REQUIRE(b && b->nutt >= 0 && b->max_states >= 1);
    if (b->nutt == 0) return BEER_OK;
    hipStream_t s = as_stream(stream);
    // LDS: two trellis columns, a chunk of back-pointers and, when they fit, the arcs
    const size_t elem = dtype == BEER_F32 ? sizeof(float) : sizeof(double);
    const size_t base = (size_t)2 * b->max_states * elem +
                        (size_t)kViterbiChunk * b->max_states * sizeof(int32_t);
    const size_t arcs = ((size_t)b->max_states + 2 + b->max_arcs) * sizeof(int32_t) +
                        (size_t)b->max_arcs * elem;
    BEER_REQUIRE(base <= 160 * 1024);
    const int arcs_in_lds = base + arcs <= 64 * 1024;
    const size_t lds = base + (arcs_in_lds ? arcs : 0);
    // four lanes per state when 512 threads suffice for that, else 256 threads
    const int vthreads = 4 * b->max_states <= kViterbiThreads ? kViterbiThreads : kHmmThreads;
    if (dtype == BEER_F32) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(viterbi_kernel<float>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, beer::kMaxDynLds);
        hipLaunchKernelGGL(viterbi_kernel<float>, dim3(b->nutt), dim3(vthreads), lds, s, *b,
                           (const float*)pc_llhs, bt_ws, path, map_pdf, arcs_in_lds);
    } else if (dtype == BEER_F64) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(viterbi_kernel<double>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, beer::kMaxDynLds);
        hipLaunchKernelGGL(viterbi_kernel<double>, dim3(b->nutt), dim3(vthreads), lds, s, *b,
                           (const double*)pc_llhs, bt_ws, path, map_pdf, arcs_in_lds);
    } else {
        return BEER_EINVAL;
    }
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

int beer_hmm_path_posteriors(int dtype, const beer_batch* b, const int64_t* path, void* gamma,
                             double* xi_sum, double* gamma0_sum, void* stream) {
    BEER_REQUIRE(b && b->nutt >= 0);
    if (b->nutt == 0) return BEER_OK;
    hipStream_t s = as_stream(stream);
    if (dtype == BEER_F32)
        hipLaunchKernelGGL(path_post_kernel<float>, dim3(b->nutt, 2), dim3(256), 0, s, *b, path,
                           (float*)gamma, xi_sum, gamma0_sum);
    else if (dtype == BEER_F64)
        hipLaunchKernelGGL(path_post_kernel<double>, dim3(b->nutt, 2), dim3(256), 0, s, *b, path,
                           (double*)gamma, xi_sum, gamma0_sum);
    else
        return BEER_EINVAL;
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

}  // extern "C"
