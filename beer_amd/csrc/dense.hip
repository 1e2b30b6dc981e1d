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

#include <mutex>
#include <unordered_map>
#include <dlfcn.h>
#include <rocblas/rocblas.h>             // types and prototypes only: the library is dlopen'ed

#include "common.h"

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
// thousands) go to rocBLAS: plain GEMMs with a long inner or outer dimension,
// where the LDS-tiled VALU kernel above runs at 25 TFLOP/s and the library's
// fp32 MFMA kernels at several times that.  The library is looked up at run time
// by its soname (inside a PyTorch process that is the copy PyTorch already
// loaded); without it, or for small / fp64 problems, the kernel above runs.
// Products are exact fp32 with fp32 accumulation; sums over frames are cut into
// kChunk-frame partial sums that are added in fp64.
// ---------------------------------------------------------------------------
constexpr int kBlasMinQ = 512;
constexpr int64_t kBlasMinT = 8192;
constexpr int kChunk = 4096;

struct RocBlas {
    decltype(&rocblas_create_handle) create = nullptr;
    decltype(&rocblas_set_stream) set_stream = nullptr;
    decltype(&rocblas_sgemm) sgemm = nullptr;
    decltype(&rocblas_sgemm_strided_batched) sgemm_sb = nullptr;
    rocblas_handle handle = nullptr;
    bool ok = false;
    RocBlas() {
        if (const char* e = getenv("BEER_NO_ROCBLAS")) { if (e[0] == '1') return; }
        void* lib = dlopen("librocblas.so.5", RTLD_NOW | RTLD_LOCAL);
        if (!lib) lib = dlopen("librocblas.so", RTLD_NOW | RTLD_LOCAL);
        if (!lib) return;
        create = reinterpret_cast<decltype(create)>(dlsym(lib, "rocblas_create_handle"));
        set_stream = reinterpret_cast<decltype(set_stream)>(dlsym(lib, "rocblas_set_stream"));
        sgemm = reinterpret_cast<decltype(sgemm)>(dlsym(lib, "rocblas_sgemm"));
        sgemm_sb = reinterpret_cast<decltype(sgemm_sb)>(
            dlsym(lib, "rocblas_sgemm_strided_batched"));
        ok = create && set_stream && sgemm && sgemm_sb;
    }
};
// One rocBLAS handle per stream, bound to it once (a handle carries its stream and
// its workspace: sharing one between streams, re-pointed per call, would make
// concurrent callers race on it).  Handles live as long as the process.
RocBlas* blas_for(hipStream_t s) {
    static RocBlas lib;                        // the library's entry points, no handle
    if (!lib.ok) return nullptr;
    static std::mutex mu;
    static std::unordered_map<hipStream_t, RocBlas*> per_stream;
    std::lock_guard<std::mutex> lock(mu);
    auto it = per_stream.find(s);
    if (it != per_stream.end()) return it->second;
    RocBlas* rb = new RocBlas(lib);
    if (lib.create(&rb->handle) != rocblas_status_success ||
        lib.set_stream(rb->handle, s) != rocblas_status_success) {
        delete rb;
        rb = nullptr;
    }
    per_stream[s] = rb;
    return rb;
}
inline bool blas_shape(int64_t T_, int Q, int K) {
    return Q >= kBlasMinQ && T_ >= kBlasMinT && T_ < (int64_t)1 << 31 && K >= 1;
}

__global__ void fill_kernel(int64_t n, float v, float* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) out[idx] = v;
}
// out[t,k] = w[t,k] * (g ? g[t] : 1) * (sr ? sr[t, k / G] : 1)
__global__ void scale_rows_kernel(int64_t T_, int K, int G, const float* __restrict__ w,
                                  const float* __restrict__ g, const float* __restrict__ sr,
                                  float* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= T_ * K) return;
    const int64_t t = idx / K;
    const int k = (int)(idx - t * K);
    float v = w[idx];
    if (g) v *= g[t];
    if (sr) v *= sr[t * (K / G) + k / G];
    out[idx] = v;
}
// acc[i] += sum_b part[b][i]  (fp64)
__global__ void add_partials_kernel(int64_t n, int nb, const float* __restrict__ part,
                                    double* __restrict__ acc) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    double sacc = 0.0;
    for (int b = 0; b < nb; ++b) sacc += (double)part[(size_t)b * n + idx];
    acc[idx] += sacc;
}

// rocBLAS is column-major: a row-major [r, c] array is its [c, r] matrix with ld = c.
// out[T,K] = stats[T,Q] @ E[K,Q]^T + base
int blas_llh(RocBlas* rb, int64_t T_, int Q, int K, const float* stats, const float* E,
             double base, float* out, hipStream_t s) {
    const int64_t n = T_ * K;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n,
                       (float)base, out);
    BEER_LAUNCH_CHECK();
    const float one = 1.f;
    return rb->sgemm(rb->handle, rocblas_operation_transpose, rocblas_operation_none, K, (int)T_,
                     Q, &one, E, Q, stats, Q, &one, out, K) == rocblas_status_success
               ? BEER_OK : BEER_EINVAL;
}
// out[T,Q] = g_t * sum_k w[t,k] E[k,q]
int blas_backward(RocBlas* rb, int64_t T_, int K, int Q, const float* w, const float* g,
                  const float* E, float* out, hipStream_t s) {
    float* wg = nullptr;
    const int64_t n = T_ * K;
    if (g) {
        if (hipMallocAsync(reinterpret_cast<void**>(&wg), (size_t)n * sizeof(float), s) !=
            hipSuccess)
            return BEER_EINVAL;
        hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                           T_, K, 1, w, g, (const float*)nullptr, wg);
    }
    const float one = 1.f, zero = 0.f;
    const rocblas_status st =
        rb->sgemm(rb->handle, rocblas_operation_none, rocblas_operation_none, Q, (int)T_, K, &one,
                  E, Q, g ? wg : w, K, &zero, out, Q);
    if (wg) (void)hipFreeAsync(wg, s);
    return st == rocblas_status_success ? BEER_OK : BEER_EINVAL;
}
// acc[K,Q] += sum_t w[t,k] sr[t, k / G] stats[t,q]
int blas_accumulate(RocBlas* rb, int64_t T_, int K, int Q, int G, const float* w,
                    const float* sr, const float* stats, double* acc, hipStream_t s) {
    const int nfull = (int)(T_ / kChunk), rest = (int)(T_ - (int64_t)nfull * kChunk);
    const int nb = nfull + (rest ? 1 : 0);
    const int64_t n = T_ * K, kq = (int64_t)K * Q;
    float* buf = nullptr;                       // [joint weights T*K (if sr)] [partials nb*K*Q]
    const size_t wj_floats = sr ? (size_t)n : 0;
    if (hipMallocAsync(reinterpret_cast<void**>(&buf),
                       (wj_floats + (size_t)nb * kq) * sizeof(float), s) != hipSuccess)
        return BEER_EINVAL;
    const float* wj = w;
    if (sr) {
        hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                           T_, K, G, w, (const float*)nullptr, sr, buf);
        wj = buf;
    }
    float* part = buf + wj_floats;
    const float one = 1.f, zero = 0.f;
    rocblas_status st = rocblas_status_success;
    // P_b [K,Q] (row-major) = wj_b^T stats_b  ==  col-major [Q,K] = stats_b^T[Q,Tb] * wj_b[Tb,K]
    if (nfull)
        st = rb->sgemm_sb(rb->handle, rocblas_operation_none, rocblas_operation_transpose, Q, K,
                          kChunk, &one, stats, Q, (rocblas_stride)kChunk * Q, wj, K,
                          (rocblas_stride)kChunk * K, &zero, part, Q, (rocblas_stride)kq, nfull);
    if (st == rocblas_status_success && rest)
        st = rb->sgemm(rb->handle, rocblas_operation_none, rocblas_operation_transpose, Q, K, rest,
                       &one, stats + (size_t)nfull * kChunk * Q, Q, wj + (size_t)nfull * kChunk * K,
                       K, &zero, part + (size_t)nfull * kq, Q);
    if (st == rocblas_status_success)
        hipLaunchKernelGGL(add_partials_kernel, dim3((unsigned)((kq + 255) / 256)), dim3(256), 0, s,
                           kq, nb, part, acc);
    (void)hipFreeAsync(buf, s);
    return st == rocblas_status_success ? BEER_OK : BEER_EINVAL;
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
    if (sizeof(T) == 4 && blas_shape(T_, Q, K))
        if (RocBlas* rb = blas_for(as_stream(stream)))
            return blas_llh(rb, T_, Q, K, (const float*)stats, (const float*)expT, base,
                            (float*)out, as_stream(stream));
    return gemm_plain<T>(T_, K, Q, (const T*)stats, Q, 1, (const T*)expT, 1, Q, (T*)out, nullptr,
                         base, as_stream(stream));
}

template <typename T>
int dense_backward_launch(int64_t T_, int K, int Q, const void* w, const void* g,
                          const void* expT, void* out, void* stream) {
    BEER_REQUIRE(T_ >= 0 && Q >= 1 && K >= 1 && w && expT && out);
    if (T_ == 0) return BEER_OK;
    // out[t,q] = g_t * sum_k w[t,k] expT[k,q]
    if (sizeof(T) == 4 && blas_shape(T_, Q, K))
        if (RocBlas* rb = blas_for(as_stream(stream)))
            return blas_backward(rb, T_, K, Q, (const float*)w, (const float*)g,
                                 (const float*)expT, (float*)out, as_stream(stream));
    return gemm_plain<T>(T_, Q, K, (const T*)w, K, 1, (const T*)expT, Q, 1, (T*)out, (const T*)g,
                         0.0, as_stream(stream));
}

template <typename T>
int dense_accumulate_launch(int64_t T_, int K, int Q, int G, const void* w, const void* sr,
                            const void* stats, double* acc, void* stream) {
    BEER_REQUIRE(T_ >= 0 && Q >= 1 && K >= 1 && G >= 1 && K % G == 0 && w && stats && acc);
    if (T_ == 0) return BEER_OK;
    // acc[k,q] += sum_t w[t,k] stats[t,q]; frames split over grid.z, fp64 atomics
    if (sizeof(T) == 4 && blas_shape(T_, Q, K))
        if (RocBlas* rb = blas_for(as_stream(stream)))
            return blas_accumulate(rb, T_, K, Q, G, (const float*)w, (const float*)sr,
                                   (const float*)stats, acc, as_stream(stream));
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
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
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

int beer_hip_has_rocblas(void) { return blas_for(nullptr) ? 1 : 0; }

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
