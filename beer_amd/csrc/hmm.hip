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

// scatter (+ per-frame expected llh): one thread per frame, serial over the
// utterance's states so that repeated pdf ids add without atomics.
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
    double mine = 0.0;
    for (int64_t t = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; t < nt;
         t += (int64_t)gridDim.y * blockDim.x) {
        T e = 0;
        T* row = sr ? sr + (f0 + t) * S_total : nullptr;
        for (int s = 0; s < S; ++s) {
            const T gv = g[t * S + s];
            if (row) row[ids[s]] += scale * gv;
            e += p[t * S + s] * gv;
        }
        if (exp_llh) exp_llh[f0 + t] = e;
        mine += (double)e;
    }
    if (utt_llh) {
        const double tot = block_sum(mine, red);
        if (threadIdx.x == 0) atomicAdd(utt_llh + u, tot);
    }
}

// ---------------------------------------------------------------------------
// forward-backward
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kHmmThreads) void fb_kernel(
    beer_batch b, const T* __restrict__ pc_llhs, double* __restrict__ alpha_ws,
    T* __restrict__ gamma, double* __restrict__ xi_sum, double* __restrict__ gamma0_sum,
    T* __restrict__ lognorm_mean) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int u = blockIdx.x, tid = threadIdx.x, nt_ = blockDim.x;
    const beer_graph g = b.graphs[b.graph_id[u]];
    const int S = g.n_states, nnz = g.n_arcs;
    const int64_t T_ = b.frame_off[u + 1] - b.frame_off[u];
    if (T_ <= 0) return;
    const T* llh = pc_llhs + b.llh_off[u];
    double* alpha = alpha_ws + b.llh_off[u];
    T* gam = gamma + b.llh_off[u];
    const T* init = (const T*)g.init;
    const T* fin = (const T*)g.final;
    const T* in_w = (const T*)g.in_w;
    const T* out_w = (const T*)g.out_w;

    double* cur = reinterpret_cast<double*>(smem);     // [S] column being read
    double* nxt = cur + S;                              // [S] column being written
    double* lb = nxt + S;                               // [S] llh_{t+1} + beta_{t+1}
    double* red = lb + S;                               // [8]
    double* xi = red + 8;                               // [nnz] (only if xi_sum)

    // ---- forward ----
    for (int j = tid; j < S; j += nt_) {
        const double a = (double)llh[j] + (double)init[j];
        cur[j] = a;
        alpha[j] = a;
    }
    __syncthreads();
    for (int64_t t = 1; t < T_; ++t) {
        for (int j = tid; j < S; j += nt_) {
            const int e0 = g.in_ptr[j], e1 = g.in_ptr[j + 1];
            double m = neg_inf();
            for (int e = e0; e < e1; ++e) {
                const double v = cur[g.in_src[e]] + (double)in_w[e];
                m = v > m ? v : m;
            }
            double lse = m;
            if (m > neg_inf() && m < __builtin_huge_val()) {
                double s = 0.0;
                for (int e = e0; e < e1; ++e) s += exp(cur[g.in_src[e]] + (double)in_w[e] - m);
                lse = m + log(s);
            }
            const double a = (double)llh[t * S + j] + lse;
            nxt[j] = a;
            alpha[t * S + j] = a;
        }
        __syncthreads();
        double* tmp = cur; cur = nxt; nxt = tmp;
    }

    // ---- backward + posteriors ----
    if (xi_sum) for (int e = tid; e < nnz; e += nt_) xi[e] = 0.0;
    // beta_{T-1} = final; `cur` holds beta_t, `lb` holds llh_{t+1}+beta_{t+1}.
    for (int j = tid; j < S; j += nt_) cur[j] = (double)fin[j];
    __syncthreads();
    double ln_acc = 0.0;
    for (int64_t t = T_ - 1; t >= 0; --t) {
        if (t < T_ - 1) {
            // beta_t(i) = lse_j(A_ij + llh_{t+1}(j) + beta_{t+1}(j))
            for (int i = tid; i < S; i += nt_) {
                const int e0 = g.out_ptr[i], e1 = g.out_ptr[i + 1];
                double m = neg_inf();
                for (int e = e0; e < e1; ++e) {
                    const double v = (double)out_w[e] + lb[g.out_dst[e]];
                    m = v > m ? v : m;
                }
                double lse = m;
                if (m > neg_inf() && m < __builtin_huge_val()) {
                    double s = 0.0;
                    for (int e = e0; e < e1; ++e) s += exp((double)out_w[e] + lb[g.out_dst[e]] - m);
                    lse = m + log(s);
                }
                cur[i] = lse;
            }
            __syncthreads();
        }
        // lognorm_t = lse_i(alpha_t(i) + beta_t(i)); gamma_t
        double m = neg_inf();
        for (int j = tid; j < S; j += nt_) {
            const double v = alpha[t * S + j] + cur[j];
            m = v > m ? v : m;
        }
        m = block_max(m, red);
        double lognorm = m;
        if (m > neg_inf() && m < __builtin_huge_val()) {
            double s = 0.0;
            for (int j = tid; j < S; j += nt_) s += exp(alpha[t * S + j] + cur[j] - m);
            s = block_sum(s, red);
            lognorm = m + log(s);
        }
        ln_acc += lognorm;
        for (int j = tid; j < S; j += nt_) {
            const double gv = exp(alpha[t * S + j] + cur[j] - lognorm);   // NaN if -inf - -inf
            gam[t * S + j] = (T)gv;
            if (t == 0 && gamma0_sum) atomicAdd(gamma0_sum + j, gv);
        }
        // xi_t(i,j) for the arcs t -> t+1 (graph.py:308-323), NaN -> 0.
        if (xi_sum && t < T_ - 1 && lognorm > neg_inf()) {
            for (int i = tid; i < S; i += nt_) {
                const double ai = alpha[t * S + i] - lognorm;
                for (int e = g.out_ptr[i]; e < g.out_ptr[i + 1]; ++e) {
                    const double v = exp(ai + (double)out_w[e] + lb[g.out_dst[e]]);
                    if (v == v) xi[e] += v;
                }
            }
        }
        __syncthreads();
        // lb <- llh_t + beta_t for the next (earlier) frame
        for (int j = tid; j < S; j += nt_) lb[j] = (double)llh[t * S + j] + cur[j];
        __syncthreads();
    }
    if (lognorm_mean && tid == 0) lognorm_mean[u] = (T)(ln_acc / (double)T_);
    if (xi_sum) {
        for (int i = tid; i < S; i += nt_)
            for (int e = g.out_ptr[i]; e < g.out_ptr[i + 1]; ++e)
                atomicAdd(xi_sum + (size_t)i * S + g.out_dst[e], xi[e]);
    }
}

// ---------------------------------------------------------------------------
// Viterbi
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kHmmThreads) void viterbi_kernel(
    beer_batch b, const T* __restrict__ pc_llhs, int32_t* __restrict__ bt_ws,
    int64_t* __restrict__ path, int map_pdf) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int u = blockIdx.x, tid = threadIdx.x, nt_ = blockDim.x;
    const int gid = b.graph_id[u];
    const beer_graph g = b.graphs[gid];
    const int S = g.n_states;
    const int64_t T_ = b.frame_off[u + 1] - b.frame_off[u];
    if (T_ <= 0) return;
    const T* llh = pc_llhs + b.llh_off[u];
    int32_t* bt = bt_ws + b.llh_off[u];
    int64_t* out = path + b.frame_off[u];
    const T* init = (const T*)g.init;
    const T* fin = (const T*)g.final;
    const T* in_w = (const T*)g.in_w;
    T* cur = reinterpret_cast<T*>(smem);
    T* nxt = cur + S;

    for (int j = tid; j < S; j += nt_) cur[j] = llh[j] + init[j];
    __syncthreads();
    for (int64_t t = 1; t < T_; ++t) {
        for (int j = tid; j < S; j += nt_) {
            // argmax_i(omega_i + A_ij): first index wins; all -inf -> 0
            // (torch.argmax over the dense column, graph.py:337-338).
            T best = ninf<T>();
            int arg = 0;
            for (int e = g.in_ptr[j]; e < g.in_ptr[j + 1]; ++e) {
                const T v = cur[g.in_src[e]] + in_w[e];
                if (v > best) { best = v; arg = g.in_src[e]; }
            }
            bt[t * S + j] = arg;
            // hypothesis[j, arg]: when no arc is finite the dense entry is
            // omega_0 + A_0j = -inf (or NaN-free -inf + finite)
            nxt[j] = llh[t * S + j] + best;
        }
        __syncthreads();
        T* tmp = cur; cur = nxt; nxt = tmp;
    }
    if (tid == 0) {
        T best = ninf<T>();
        int arg = 0;
        for (int j = 0; j < S; ++j) {
            const T v = cur[j] + fin[j];
            if (v > best) { best = v; arg = j; }
        }
        const int32_t* ids = b.pdf_ids + b.pdf_off[gid];
        int s = arg;
        for (int64_t t = T_ - 1; t >= 0; --t) {
            out[t] = map_pdf ? (int64_t)ids[s] : (int64_t)s;
            if (t > 0) s = bt[t * S + s];
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
    hipLaunchKernelGGL(scatter_kernel<T>, dim3(b->nutt, 1), dim3(256), 0, as_stream(stream), *b,
                       S_total, (const T*)pc, (const T*)gamma, (T)scale, (T*)sr, (T*)exp_llh,
                       utt_llh);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

}  // namespace

extern "C" {

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
                              double* alpha_ws, void* gamma, double* xi_sum, double* gamma0_sum,
                              void* lognorm_mean, void* stream) {
    BEER_REQUIRE(b && b->nutt >= 0 && b->max_states >= 1);
    if (b->nutt == 0) return BEER_OK;
    const size_t lds = ((size_t)3 * b->max_states + 8 + (xi_sum ? (size_t)b->max_arcs : 0)) *
                       sizeof(double);
    BEER_REQUIRE(lds <= 160 * 1024);
    hipStream_t s = as_stream(stream);
    if (dtype == BEER_F32) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fb_kernel<float>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(fb_kernel<float>, dim3(b->nutt), dim3(kHmmThreads), lds, s, *b,
                           (const float*)pc_llhs, alpha_ws, (float*)gamma, xi_sum,
                           gamma0_sum, (float*)lognorm_mean);
    } else if (dtype == BEER_F64) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fb_kernel<double>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(fb_kernel<double>, dim3(b->nutt), dim3(kHmmThreads), lds, s, *b,
                           (const double*)pc_llhs, alpha_ws, (double*)gamma, xi_sum,
                           gamma0_sum, (double*)lognorm_mean);
    } else {
        return BEER_EINVAL;
    }
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

int beer_hmm_viterbi(int dtype, const beer_batch* b, const void* pc_llhs, int32_t* bt_ws,
                     int64_t* path, int map_pdf, void* stream) {
    BEER_REQUIRE(b && b->nutt >= 0 && b->max_states >= 1);
    if (b->nutt == 0) return BEER_OK;
    hipStream_t s = as_stream(stream);
    if (dtype == BEER_F32) {
        const size_t lds = (size_t)2 * b->max_states * sizeof(float);
        BEER_REQUIRE(lds <= 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(viterbi_kernel<float>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(viterbi_kernel<float>, dim3(b->nutt), dim3(kHmmThreads), lds, s, *b,
                           (const float*)pc_llhs, bt_ws, path, map_pdf);
    } else if (dtype == BEER_F64) {
        const size_t lds = (size_t)2 * b->max_states * sizeof(double);
        BEER_REQUIRE(lds <= 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(viterbi_kernel<double>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(viterbi_kernel<double>, dim3(b->nutt), dim3(kHmmThreads), lds, s, *b,
                           (const double*)pc_llhs, bt_ws, path, map_pdf);
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
