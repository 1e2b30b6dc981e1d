// Exponential-family parameter kernels: E_q[T], log-normalisers, natural <->
// standard parameter maps, KL divergence and the natural-gradient step for
// the Normal-Wishart / Normal-Gamma / isotropic Normal-Gamma / Dirichlet /
// Gamma posteriors.  They run once per VB iteration over K distributions, so
// they are written for accuracy (all arithmetic in fp64 whatever the storage
// type; D x D factorisations stay in LDS) rather than for throughput.
//
// Reference restated: beer/dists/{normalwishart,normalgamma,isonormalgamma,
// dirichlet,gamma,basedist}.py and beer/models/parameters.py:134-141.

#include "common.h"

using namespace beer;

namespace {

constexpr int kNwThreads = 1024;    // launch bound of the one-workgroup-per-matrix kernels
// They are latency chains of D elimination steps with D*D / threads entry updates
// each: with few matrices (one per CU or less) 1024 threads are 1.5-2x faster than
// 256 (K = 256, D = 40: 68 -> 36 us); with many, smaller workgroups pack better.
inline int nw_threads(int K, int D) {
    return K <= 512 && D >= 16 ? 1024 : 256;
}
constexpr int kMaxFullDim = 128;   // D*D fp64 must fit one CU's 160 KiB LDS

// ---------------------------------------------------------------------------
// In-LDS SPD factorisation helpers (one workgroup per matrix, A is D x D
// row-major fp64 in LDS).
// ---------------------------------------------------------------------------

// (spd_inverse: common.h -- also used when the E-step packs its parameters)

// ---------------------------------------------------------------------------
// Normal-Wishart
// ---------------------------------------------------------------------------

template <typename T>
__global__ __launch_bounds__(kNwThreads) void nw_expected_stats_kernel(
    int D, const T* __restrict__ mean, const T* __restrict__ scale,
    const T* __restrict__ W, const T* __restrict__ dof, T* __restrict__ out,
    T* __restrict__ lnorm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* A = reinterpret_cast<double*>(smem);       // D*D
    double* m = A + D * D;                              // D
    double* pm = m + D;                                 // D : nu W m
    double* red = pm + D;                               // 8
    const int k = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int Q = D * D + D + 2;
    const double nu = (double)dof[k], kappa = (double)scale[k];
    const T* Wk = W + (size_t)k * D * D;
    T* o = out + (size_t)k * Q;
    for (int i = tid; i < D * D; i += nt) {
        const double w = (double)Wk[i];
        A[i] = w;
        o[D + i] = (T)(nu * w);
    }
    for (int i = tid; i < D; i += nt) m[i] = (double)mean[(size_t)k * D + i];
    __syncthreads();
    for (int i = tid; i < D; i += nt) {
        double s = 0.0;
        for (int j = 0; j < D; ++j) s += A[i * D + j] * m[j];
        pm[i] = nu * s;
        o[i] = (T)pm[i];
    }
    __syncthreads();
    double part = 0.0;
    for (int i = tid; i < D; i += nt) part += pm[i] * m[i];
    const double tr = block_sum(part, red);
    double dg = 0.0;
    for (int i = tid; i < D; i += nt) dg += digamma(0.5 * (nu + 1.0 - (double)(i + 1)));
    const double dgs = block_sum(dg, red);
    double lgs = 0.0;
    if (lnorm) {                   // the log-normaliser shares the factorisation
        double lg = 0.0;
        for (int i = tid; i < D; i += nt) lg += lgamma(0.5 * (nu + 1.0 - (double)(i + 1)));
        lgs = block_sum(lg, red);
    }
    const double logdet = spd_inverse(A, D, red + 8, false);
    if (tid == 0) {
        o[D + D * D] = (T)((double)D / kappa + tr);
        o[D + D * D + 1] = (T)(dgs + (double)D * kLog2 + logdet);
        if (lnorm) {
            const double d = (double)D;
            lnorm[k] = (T)(0.5 * nu * logdet + 0.5 * nu * d * kLog2 +
                           0.25 * d * (d - 1.0) * kLogPi + lgs - 0.5 * d * log(kappa) +
                           0.5 * d * kLog2Pi);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(kNwThreads) void nw_log_norm_kernel(
    int D, const T* __restrict__ scale, const T* __restrict__ W,
    const T* __restrict__ dof, T* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* A = reinterpret_cast<double*>(smem);
    double* red = A + D * D;
    const int k = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const double nu = (double)dof[k], kappa = (double)scale[k];
    const T* Wk = W + (size_t)k * D * D;
    for (int i = tid; i < D * D; i += nt) A[i] = (double)Wk[i];
    double lg = 0.0;
    for (int i = tid; i < D; i += nt) lg += lgamma(0.5 * (nu + 1.0 - (double)(i + 1)));
    const double lgs = block_sum(lg, red);
    const double logdet = spd_inverse(A, D, red + 8, false);
    if (tid == 0) {
        const double d = (double)D;
        out[k] = (T)(0.5 * nu * logdet + 0.5 * nu * d * kLog2 +
                     0.25 * d * (d - 1.0) * kLogPi + lgs - 0.5 * d * log(kappa) +
                     0.5 * d * kLog2Pi);
    }
}

template <typename T>
__global__ __launch_bounds__(kNwThreads) void nw_natural_kernel(
    int D, const T* __restrict__ mean, const T* __restrict__ scale,
    const T* __restrict__ W, const T* __restrict__ dof, T* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* A = reinterpret_cast<double*>(smem);
    double* m = A + D * D;
    double* red = m + D;
    const int k = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int Q = D * D + D + 2;
    const double nu = (double)dof[k], kappa = (double)scale[k];
    const T* Wk = W + (size_t)k * D * D;
    T* o = out + (size_t)k * Q;
    for (int idx = tid; idx < D * D; idx += nt) {
        const int i = idx / D, j = idx % D;
        // symmetrise the (numerically almost symmetric) scale matrix
        A[idx] = 0.5 * ((double)Wk[i * D + j] + (double)Wk[j * D + i]);
    }
    for (int i = tid; i < D; i += nt) {
        m[i] = (double)mean[(size_t)k * D + i];
        o[i] = (T)(kappa * m[i]);
    }
    spd_inverse(A, D, red + 8, true);
    for (int idx = tid; idx < D * D; idx += nt) {
        const int i = idx / D, j = idx % D;
        // (the elimination leaves A^-1 symmetric up to rounding: average)
        o[D + idx] = (T)(-0.5 * (0.5 * (A[i * D + j] + A[j * D + i]) + kappa * m[i] * m[j]));
    }
    if (tid == 0) {
        o[D + D * D] = (T)(-0.5 * kappa);
        o[D + D * D + 1] = (T)(0.5 * (nu - (double)D));
    }
}

template <typename T>
__global__ __launch_bounds__(kNwThreads) void nw_from_natural_kernel(
    int D, const T* __restrict__ eta, T* __restrict__ mean, T* __restrict__ scale,
    T* __restrict__ W, T* __restrict__ dof) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* A = reinterpret_cast<double*>(smem);
    double* m = A + D * D;
    double* red = m + D;
    const int k = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int Q = D * D + D + 2;
    const T* e = eta + (size_t)k * Q;
    const double kappa = -2.0 * (double)e[D + D * D];
    for (int i = tid; i < D; i += nt) {
        m[i] = (double)e[i] / kappa;
        mean[(size_t)k * D + i] = (T)m[i];
    }
    __syncthreads();
    for (int idx = tid; idx < D * D; idx += nt) {
        const int i = idx / D, j = idx % D;
        const double b = 0.5 * ((double)e[D + i * D + j] + (double)e[D + j * D + i]);
        A[idx] = -2.0 * b - kappa * m[i] * m[j];
    }
    spd_inverse(A, D, red + 8, true);
    T* Wk = W + (size_t)k * D * D;
    for (int idx = tid; idx < D * D; idx += nt) {
        const int i = idx / D, j = idx - i * D;
        Wk[idx] = (T)(0.5 * (A[i * D + j] + A[j * D + i]));
    }
    if (tid == 0) {
        scale[k] = (T)kappa;
        dof[k] = (T)(2.0 * (double)e[D + D * D + 1] + (double)D);
    }
}

// The M-step of a Normal-Wishart posterior in ONE launch: natural parameters ->
// standard parameters (nw_from_natural_kernel) AND, from the same elimination, what
// the next iteration asks of the new posterior -- E[T], the log-normaliser
// (nw_expected_stats_kernel) and the moments of its expected Gaussian (mean, E[Lambda]^-1
// = W^-1 / nu: the matrix that is being inverted here anyway).  The inverse of W^-1
// yields W and log|W^-1| = -log|W| together: one factorisation where the separate
// calls make two.  E[T] and the log-normaliser are computed from the parameters as
// they are STORED (rounded to T), like the separate calls, except for log|W| (taken
// from the unrounded elimination; the difference is below T's rounding of the
// result).
template <typename T>
__global__ __launch_bounds__(kNwThreads) void nw_update_kernel(
    int D, const T* __restrict__ eta, T* __restrict__ mean, T* __restrict__ scale,
    T* __restrict__ W, T* __restrict__ dof, T* __restrict__ out, T* __restrict__ lnorm,
    T* __restrict__ moments) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* A = reinterpret_cast<double*>(smem);       // D*D
    double* m = A + D * D;                              // D
    double* pm = m + D;                                 // D : nu W m
    double* red = pm + D;                               // 8 (+ 2 D of spd_inverse behind)
    const int k = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int Q = D * D + D + 2;
    const T* e = eta + (size_t)k * Q;
    const T kappa_t = (T)(-2.0 * (double)e[D + D * D]);
    const T nu_t = (T)(2.0 * (double)e[D + D * D + 1] + (double)D);
    const double kappa = (double)kappa_t, nu = (double)nu_t;
    const double kappa_u = -2.0 * (double)e[D + D * D];
    for (int i = tid; i < D; i += nt) {
        const double mu = (double)e[i] / kappa_u;
        const T mt = (T)mu;
        mean[(size_t)k * D + i] = mt;
        m[i] = mu;
        if (moments) moments[(size_t)k * (D + D * D) + i] = mt;
    }
    __syncthreads();
    for (int idx = tid; idx < D * D; idx += nt) {
        const int i = idx / D, j = idx % D;
        const double b = 0.5 * ((double)e[D + i * D + j] + (double)e[D + j * D + i]);
        const double winv = -2.0 * b - kappa_u * m[i] * m[j];
        A[idx] = winv;
        if (moments) moments[(size_t)k * (D + D * D) + D + idx] = (T)(winv / nu);
    }
    const double logdet = -spd_inverse(A, D, red + 8, true);        // log|W|
    T* Wk = W + (size_t)k * D * D;
    T* o = out + (size_t)k * Q;
    __syncthreads();
    for (int i = tid; i < D; i += nt) m[i] = (double)mean[(size_t)k * D + i];   // as stored
    __syncthreads();
    for (int idx = tid; idx < D * D; idx += nt) {
        const int i = idx / D, j = idx - i * D;
        const T w = (T)(0.5 * (A[i * D + j] + A[j * D + i]));
        Wk[idx] = w;
        o[D + idx] = (T)(nu * (double)w);
    }
    __syncthreads();
    for (int i = tid; i < D; i += nt) {
        double sacc = 0.0;
        for (int j = 0; j < D; ++j) sacc += (double)Wk[i * D + j] * m[j];
        pm[i] = nu * sacc;
        o[i] = (T)pm[i];
    }
    __syncthreads();
    double part = 0.0;
    for (int i = tid; i < D; i += nt) part += pm[i] * m[i];
    const double tr = block_sum(part, red);
    double dg = 0.0, lg = 0.0;
    for (int i = tid; i < D; i += nt) {
        dg += digamma(0.5 * (nu + 1.0 - (double)(i + 1)));
        lg += lgamma(0.5 * (nu + 1.0 - (double)(i + 1)));
    }
    const double dgs = block_sum(dg, red);
    const double lgs = block_sum(lg, red);
    if (tid == 0) {
        scale[k] = kappa_t;
        dof[k] = nu_t;
        o[D + D * D] = (T)((double)D / kappa + tr);
        o[D + D * D + 1] = (T)(dgs + (double)D * kLog2 + logdet);
        const double d = (double)D;
        lnorm[k] = (T)(0.5 * nu * logdet + 0.5 * nu * d * kLog2 +
                       0.25 * d * (d - 1.0) * kLogPi + lgs - 0.5 * d * log(kappa) +
                       0.5 * d * kLog2Pi);
    }
}

// ---------------------------------------------------------------------------
// Normal-Gamma (diag) and isotropic Normal-Gamma: one thread per pdf.
// ---------------------------------------------------------------------------

template <typename T, bool ISO>
__global__ void ng_expected_stats_kernel(int K, int D, const T* mean, const T* scale,
                                         const T* shape, const T* rates, T* out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const int Q = ISO ? D + 3 : 2 * D + 2;
    const double a = (double)shape[k], kappa = (double)scale[k];
    T* o = out + (size_t)k * Q;
    double pqm = 0.0, logdet = 0.0;
    const double psi = digamma(a);
    if (ISO) {
        const double b = (double)rates[k], prec = a / b;
        double m2 = 0.0;
        for (int d = 0; d < D; ++d) {
            const double m = (double)mean[(size_t)k * D + d];
            o[d] = (T)(prec * m);
            m2 += m * m;
        }
        o[D] = (T)prec;
        pqm = prec * m2;
        logdet = psi - log(b);
    } else {
        for (int d = 0; d < D; ++d) {
            const double m = (double)mean[(size_t)k * D + d];
            const double b = (double)rates[(size_t)k * D + d], prec = a / b;
            o[d] = (T)(prec * m);
            o[D + d] = (T)prec;
            pqm += prec * m * m;
            logdet += psi - log(b);
        }
    }
    o[Q - 2] = (T)(pqm + (double)D / kappa);
    o[Q - 1] = (T)logdet;
}

template <typename T, bool ISO>
__global__ void ng_log_norm_kernel(int K, int D, const T* scale, const T* shape,
                                   const T* rates, T* out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const double a = (double)shape[k], kappa = (double)scale[k];
    double r;
    if (ISO) {
        r = lgamma(a) - a * log((double)rates[k]) - 0.5 * (double)D * log(kappa);
    } else {
        double sl = 0.0;
        for (int d = 0; d < D; ++d) sl += log((double)rates[(size_t)k * D + d]);
        r = (double)D * lgamma(a) - a * sl - 0.5 * (double)D * log(kappa);
    }
    out[k] = (T)r;
}

template <typename T, bool ISO>
__global__ void ng_natural_kernel(int K, int D, const T* mean, const T* scale,
                                  const T* shape, const T* rates, T* out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const int Q = ISO ? D + 3 : 2 * D + 2;
    const double a = (double)shape[k], kappa = (double)scale[k];
    T* o = out + (size_t)k * Q;
    double m2 = 0.0;
    for (int d = 0; d < D; ++d) {
        const double m = (double)mean[(size_t)k * D + d];
        o[d] = (T)(kappa * m);
        if (!ISO) o[D + d] = (T)(-0.5 * kappa * m * m - (double)rates[(size_t)k * D + d]);
        m2 += m * m;
    }
    if (ISO) {
        o[D] = (T)(-0.5 * kappa * m2 - (double)rates[k]);
        o[Q - 1] = (T)(a - 1.0 + 0.5 * (double)D);
    } else {
        o[Q - 1] = (T)(a - 0.5);
    }
    o[Q - 2] = (T)(-0.5 * kappa);
}

template <typename T, bool ISO>
__global__ void ng_from_natural_kernel(int K, int D, const T* eta, T* mean, T* scale,
                                       T* shape, T* rates) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const int Q = ISO ? D + 3 : 2 * D + 2;
    const T* e = eta + (size_t)k * Q;
    const double kappa = -2.0 * (double)e[Q - 2];
    double m2 = 0.0;
    for (int d = 0; d < D; ++d) {
        const double m = (double)e[d] / kappa;
        mean[(size_t)k * D + d] = (T)m;
        if (!ISO) rates[(size_t)k * D + d] = (T)(-(double)e[D + d] - 0.5 * kappa * m * m);
        m2 += m * m;
    }
    scale[k] = (T)kappa;
    if (ISO) {
        shape[k] = (T)((double)e[Q - 1] + 1.0 - 0.5 * (double)D);
        rates[k] = (T)(-(double)e[D] - 0.5 * kappa * m2);
    } else {
        shape[k] = (T)((double)e[Q - 1] + 0.5);
    }
}

// ---------------------------------------------------------------------------
// Dirichlet / Gamma: one thread per pdf.
// ---------------------------------------------------------------------------

// One wave per pdf (row); lanes stride over the G categories.
// mode 0: E[T]; 1: natural; 2: from_natural; 3: log weights (eye @ E[T]).
template <typename T, int MODE>
__global__ __launch_bounds__(64) void dirichlet_kernel(int S, int G, const T* in, T* out) {
    const int s = blockIdx.x, lane = threadIdx.x;
    const T* c = in + (size_t)s * G;
    T* o = out + (size_t)s * G;
    if (MODE == 0 || MODE == 3) {
        double part = 0.0;
        for (int g = lane; g < G; g += 64) part += (double)c[g];
        const double tot = wave_sum(part);
        const double psi_last = digamma((double)c[G - 1]);
        const double last = psi_last - digamma(tot);
        for (int g = lane; g < G - 1; g += 64) {
            const double e = digamma((double)c[g]) - psi_last;
            o[g] = (T)(MODE == 0 ? e : e + last);
        }
        if (lane == 0) o[G - 1] = (T)last;
    } else if (MODE == 1) {
        double part = 0.0;
        for (int g = lane; g < G; g += 64) {
            part += (double)c[g] - 1.0;
            if (g < G - 1) o[g] = (T)((double)c[g] - 1.0);
        }
        const double tot = wave_sum(part);
        if (lane == 0) o[G - 1] = (T)tot;
    } else {
        double part = 0.0;
        for (int g = lane; g < G - 1; g += 64) {
            part += (double)c[g];
            o[g] = (T)((double)c[g] + 1.0);
        }
        const double tot = wave_sum(part);
        if (lane == 0) o[G - 1] = (T)((double)c[G - 1] - tot + 1.0);
    }
}

template <typename T>
__global__ __launch_bounds__(64) void dirichlet_log_norm_kernel(int S, int G, const T* conc,
                                                                T* out) {
    const int s = blockIdx.x, lane = threadIdx.x;
    double tot = 0.0, lg = 0.0;
    for (int g = lane; g < G; g += 64) {
        const double c = (double)conc[(size_t)s * G + g];
        tot += c;
        lg += lgamma(c);
    }
    tot = wave_sum(tot);
    lg = wave_sum(lg);
    if (lane == 0) out[s] = (T)(lg - lgamma(tot));
}

// mode 0: E[T] [2n]; 1: natural [2n]; 2: from natural; 3: log_norm [1].
template <typename T, int MODE>
__global__ void gamma_kernel(int n, const T* a, const T* b, T* out, T* out2) {
    if (MODE == 3) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            double r = 0.0;
            for (int i = 0; i < n; ++i)
                r += lgamma((double)a[i]) - (double)a[i] * log((double)b[i]);
            out[0] = (T)r;
        }
        return;
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (MODE == 0) {
        out[i] = (T)((double)a[i] / (double)b[i]);
        out[n + i] = (T)(digamma((double)a[i]) - log((double)b[i]));
    } else if (MODE == 1) {
        out[i] = (T)(-(double)b[i]);
        out[n + i] = (T)((double)a[i] - 1.0);
    } else {             // a = eta [2n]; out = shape, out2 = rate
        out[i] = (T)((double)a[n + i] + 1.0);
        out2[i] = (T)(-(double)a[i]);
    }
}

// ---------------------------------------------------------------------------
// KL, natural-gradient step
// ---------------------------------------------------------------------------

template <typename T>
__global__ __launch_bounds__(64) void kl_kernel(int Q, const T* es, const T* eq,
                                                const T* ep, const T* lq,
                                                const T* lp, T* out) {
    const int k = blockIdx.x;
    double s = 0.0;
    for (int i = threadIdx.x; i < Q; i += 64) {
        const size_t j = (size_t)k * Q + i;
        s += (double)es[j] * ((double)ep[j] - (double)eq[j]);
    }
    s = wave_sum(s);
    if (threadIdx.x == 0) out[k] = (T)((double)lp[k] - (double)lq[k] - s);
}

template <typename T>
__global__ void nat_grad_kernel(int64_t n, const T* ep, const T* eq, const T* st,
                                double lr, T* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double q = (double)eq[i];
    out[i] = (T)(q + lr * ((double)ep[i] + (double)st[i] - q));
}

template <typename T>
__global__ void suffstats_kernel(int cov, int64_t T_, int D, const T* X, T* out) {
    const int Q = stats_dim(cov, D);
    const int64_t total = T_ * Q;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = idx / Q;
        const int q = (int)(idx % Q);
        const T* x = X + t * D;
        T v;
        if (q < D) {
            v = x[q];
        } else if (q >= Q - 2) {
            v = (q == Q - 2) ? (T)-0.5 : (cov == BEER_ISO ? (T)(0.5 * D) : (T)0.5);
        } else if (cov == BEER_FULL) {
            const int r = q - D;
            v = (T)-0.5 * (x[r / D] * x[r % D]);
        } else if (cov == BEER_DIAG) {
            v = (T)-0.5 * (x[q - D] * x[q - D]);
        } else {
            T s = 0;
            for (int d = 0; d < D; ++d) s += x[d] * x[d];
            v = (T)-0.5 * s;
        }
        out[idx] = v;
    }
}

inline int blocks_for(int64_t n, int bs) { return (int)((n + bs - 1) / bs); }

// ---- host-side typed launchers --------------------------------------------

template <typename T>
int nw_launch(int which, int K, int D, const void* mean, const void* scale,
              const void* W, const void* dof, void* out, void* stream, void* lnorm = nullptr) {
    BEER_REQUIRE(K >= 0 && D >= 1 && D <= kMaxFullDim);
    if (K == 0) return BEER_OK;
    const size_t lds = ((size_t)D * D + 4 * D + 16) * sizeof(double);
    hipStream_t s = as_stream(stream);
    if (which == 0) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nw_expected_stats_kernel<T>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, beer::kMaxDynLds);
        hipLaunchKernelGGL(nw_expected_stats_kernel<T>, dim3(K), dim3(nw_threads(K, D)), lds, s, D,
                           (const T*)mean, (const T*)scale, (const T*)W, (const T*)dof, (T*)out,
                           (T*)lnorm);
    } else if (which == 1) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nw_log_norm_kernel<T>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, beer::kMaxDynLds);
        hipLaunchKernelGGL(nw_log_norm_kernel<T>, dim3(K), dim3(nw_threads(K, D)), lds, s, D,
                           (const T*)scale, (const T*)W, (const T*)dof, (T*)out);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nw_natural_kernel<T>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, beer::kMaxDynLds);
        hipLaunchKernelGGL(nw_natural_kernel<T>, dim3(K), dim3(nw_threads(K, D)), lds, s, D,
                           (const T*)mean, (const T*)scale, (const T*)W, (const T*)dof, (T*)out);
    }
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

template <typename T>
int nw_from_natural_launch(int K, int D, const void* eta, void* mean, void* scale,
                           void* W, void* dof, void* stream) {
    BEER_REQUIRE(K >= 0 && D >= 1 && D <= kMaxFullDim);
    if (K == 0) return BEER_OK;
    const size_t lds = ((size_t)D * D + 4 * D + 16) * sizeof(double);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nw_from_natural_kernel<T>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, beer::kMaxDynLds);
    hipLaunchKernelGGL(nw_from_natural_kernel<T>, dim3(K), dim3(nw_threads(K, D)), lds,
                       as_stream(stream), D, (const T*)eta, (T*)mean, (T*)scale, (T*)W, (T*)dof);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

template <typename T>
int nw_update_launch(int K, int D, const void* eta, void* mean, void* scale, void* W, void* dof,
                     void* out, void* lnorm, void* moments, void* stream) {
    BEER_REQUIRE(K >= 0 && D >= 1 && D <= kMaxFullDim);
    if (K == 0) return BEER_OK;
    BEER_REQUIRE(eta && mean && scale && W && dof && out && lnorm);
    const size_t lds = ((size_t)D * D + 4 * D + 16) * sizeof(double);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nw_update_kernel<T>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, beer::kMaxDynLds);
    hipLaunchKernelGGL(nw_update_kernel<T>, dim3(K), dim3(nw_threads(K, D)), lds,
                       as_stream(stream), D, (const T*)eta, (T*)mean, (T*)scale, (T*)W, (T*)dof,
                       (T*)out, (T*)lnorm, (T*)moments);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

template <typename T, bool ISO>
int ng_launch(int which, int K, int D, const void* mean, const void* scale,
              const void* shape, const void* rates, void* out, void* stream) {
    BEER_REQUIRE(K >= 0 && D >= 1);
    if (K == 0) return BEER_OK;
    hipStream_t s = as_stream(stream);
    const dim3 g(blocks_for(K, 64)), b(64);
    if (which == 0)
        hipLaunchKernelGGL((ng_expected_stats_kernel<T, ISO>), g, b, 0, s, K, D, (const T*)mean,
                           (const T*)scale, (const T*)shape, (const T*)rates, (T*)out);
    else if (which == 1)
        hipLaunchKernelGGL((ng_log_norm_kernel<T, ISO>), g, b, 0, s, K, D, (const T*)scale,
                           (const T*)shape, (const T*)rates, (T*)out);
    else
        hipLaunchKernelGGL((ng_natural_kernel<T, ISO>), g, b, 0, s, K, D, (const T*)mean,
                           (const T*)scale, (const T*)shape, (const T*)rates, (T*)out);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

template <typename T, bool ISO>
int ng_from_natural_launch(int K, int D, const void* eta, void* mean, void* scale,
                           void* shape, void* rates, void* stream) {
    BEER_REQUIRE(K >= 0 && D >= 1);
    if (K == 0) return BEER_OK;
    hipLaunchKernelGGL((ng_from_natural_kernel<T, ISO>), dim3(blocks_for(K, 64)), dim3(64), 0,
                       as_stream(stream), K, D, (const T*)eta, (T*)mean, (T*)scale, (T*)shape,
                       (T*)rates);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

template <typename T>
int dirichlet_launch(int mode, int S, int G, const void* in, void* out, void* stream) {
    BEER_REQUIRE(S >= 0 && G >= 1);
    if (S == 0) return BEER_OK;
    hipStream_t s = as_stream(stream);
    const dim3 g(S), b(64);
    switch (mode) {
        case 0: hipLaunchKernelGGL((dirichlet_kernel<T, 0>), g, b, 0, s, S, G, (const T*)in, (T*)out); break;
        case 1: hipLaunchKernelGGL((dirichlet_kernel<T, 1>), g, b, 0, s, S, G, (const T*)in, (T*)out); break;
        case 2: hipLaunchKernelGGL((dirichlet_kernel<T, 2>), g, b, 0, s, S, G, (const T*)in, (T*)out); break;
        case 3: hipLaunchKernelGGL((dirichlet_kernel<T, 3>), g, b, 0, s, S, G, (const T*)in, (T*)out); break;
        default: hipLaunchKernelGGL(dirichlet_log_norm_kernel<T>, g, b, 0, s, S, G, (const T*)in, (T*)out);
    }
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

template <typename T>
int gamma_launch(int mode, int n, const void* a, const void* b, void* out, void* out2,
                 void* stream) {
    BEER_REQUIRE(n >= 1);
    hipStream_t s = as_stream(stream);
    const dim3 g(blocks_for(n, 64)), bl(64);
    switch (mode) {
        case 0: hipLaunchKernelGGL((gamma_kernel<T, 0>), g, bl, 0, s, n, (const T*)a, (const T*)b, (T*)out, (T*)out2); break;
        case 1: hipLaunchKernelGGL((gamma_kernel<T, 1>), g, bl, 0, s, n, (const T*)a, (const T*)b, (T*)out, (T*)out2); break;
        case 2: hipLaunchKernelGGL((gamma_kernel<T, 2>), g, bl, 0, s, n, (const T*)a, (const T*)b, (T*)out, (T*)out2); break;
        default: hipLaunchKernelGGL((gamma_kernel<T, 3>), g, bl, 0, s, n, (const T*)a, (const T*)b, (T*)out, (T*)out2);
    }
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

template <typename T>
int kl_launch(int K, int Q, const void* es, const void* eq, const void* ep, const void* lq,
              const void* lp, void* out, void* stream) {
    BEER_REQUIRE(K >= 0 && Q >= 1);
    if (K == 0) return BEER_OK;
    hipLaunchKernelGGL(kl_kernel<T>, dim3(K), dim3(64), 0, as_stream(stream), Q, (const T*)es,
                       (const T*)eq, (const T*)ep, (const T*)lq, (const T*)lp, (T*)out);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

template <typename T>
int nat_grad_launch(int64_t n, const void* ep, const void* eq, const void* st, double lr,
                    void* out, void* stream) {
    BEER_REQUIRE(n >= 0);
    if (n == 0) return BEER_OK;
    hipLaunchKernelGGL(nat_grad_kernel<T>, dim3(blocks_for(n, 256)), dim3(256), 0,
                       as_stream(stream), n, (const T*)ep, (const T*)eq, (const T*)st, lr, (T*)out);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

template <typename T>
int suffstats_launch(int cov, int64_t T_, int D, const void* X, void* out, void* stream) {
    BEER_REQUIRE(T_ >= 0 && D >= 1 && cov >= 0 && cov <= 2);
    if (T_ == 0) return BEER_OK;
    const int64_t total = T_ * stats_dim(cov, D);
    const int blocks = (int)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256);
    hipLaunchKernelGGL(suffstats_kernel<T>, dim3(blocks), dim3(256), 0, as_stream(stream), cov, T_,
                       D, (const T*)X, (T*)out);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

}  // namespace

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------

// ---------------------------------------------------------------------------
// Truncated stick-breaking categorical (beer/models/categorical.py:106-131): the
// bookkeeping between the phone counts and the Dirichlet pairs of the sticks.
// P <= 1024 sticks: one workgroup, every thread ranks its own stick against all
// others (stable: equal counts keep their index order, as the oracle's
// argsort(-counts, kind='stable')).
// ---------------------------------------------------------------------------
namespace {

constexpr int kSbMax = 1024;

template <typename T>
__device__ inline int sb_rank(const T* v, int P, int i) {          // position in descending order
    int r = 0;
    const T mine = v[i];
    for (int j = 0; j < P; ++j) r += (v[j] > mine) || (v[j] == mine && j < i);
    return r;
}

// counts [P] -> ordering [P] (stick -> category), stats [P,2] = (count_i, count_i +
// sum of the counts ranked after i) in the categories' own order
template <typename T>
__global__ void sb_transform_kernel(int P, const T* __restrict__ counts,
                                    int64_t* __restrict__ ordering, T* __restrict__ stats) {
    __shared__ T v[kSbMax];
    __shared__ int rank[kSbMax], ord[kSbMax];
    for (int i = threadIdx.x; i < P; i += blockDim.x) v[i] = counts[i];
    __syncthreads();
    for (int i = threadIdx.x; i < P; i += blockDim.x) {
        rank[i] = sb_rank(v, P, i);
        ord[rank[i]] = i;
        ordering[rank[i]] = i;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < P; i += blockDim.x) {
        // sum of the later sticks, added from the last one up (the reference's flipped
        // cumulative sum)
        double tail = 0.0;
        for (int r = P - 1; r > rank[i]; --r) tail += (double)v[ord[r]];
        stats[2 * i] = v[i];
        stats[2 * i + 1] = (T)(tail + (double)v[i]);
    }
}

// concentrations [P,2], ordering [P] -> E[ln pi_i] [P] (categories' order) and
// sum_i E[ln(1 - v_i)]
template <typename T>
__global__ void sb_log_weights_kernel(int P, const T* __restrict__ conc,
                                      const int64_t* __restrict__ ordering,
                                      T* __restrict__ log_w, T* __restrict__ log_1_v_sum) {
    __shared__ double l1v[kSbMax];          // E[ln(1 - v)] by stick
    __shared__ double lv[kSbMax];
    for (int r = threadIdx.x; r < P; r += blockDim.x) {
        const int i = (int)ordering[r];
        const double a = (double)conc[2 * i], b = (double)conc[2 * i + 1], sd = digamma(a + b);
        lv[r] = digamma(a) - sd;
        l1v[r] = digamma(b) - sd;
    }
    __syncthreads();
    for (int r = threadIdx.x; r < P; r += blockDim.x) {
        double acc = lv[r];
        for (int q = 0; q < r; ++q) acc += l1v[q];
        log_w[ordering[r]] = (T)acc;
    }
    if (threadIdx.x == 0 && log_1_v_sum) {
        double tot = 0.0;
        for (int r = 0; r < P; ++r) tot += l1v[r];
        *log_1_v_sum = (T)tot;
    }
}

// ... and for truncations beyond what a workgroup's static LDS arrays hold (the reference
// takes any truncation, beer/models/categorical.py:84-86): the same arithmetic in the same
// order on the global arrays themselves -- `ordering` (an output) is the stick -> category
// table of the second phase, ranks and the sticks' E[ln(1 - v)] are recomputed where they are
// needed.  O(P^2 / 1024) per thread: a few ms at P = 10 000, nobody's hot path.
template <typename T>
__global__ void sb_transform_big_kernel(int P, const T* __restrict__ counts,
                                        int64_t* __restrict__ ordering, T* __restrict__ stats) {
    for (int i = threadIdx.x; i < P; i += blockDim.x) ordering[sb_rank(counts, P, i)] = i;
    __syncthreads();                                  // (one workgroup: its global writes are visible)
    for (int i = threadIdx.x; i < P; i += blockDim.x) {
        const int rank = sb_rank(counts, P, i);
        double tail = 0.0;
        for (int r = P - 1; r > rank; --r) tail += (double)counts[ordering[r]];
        stats[2 * i] = counts[i];
        stats[2 * i + 1] = (T)(tail + (double)counts[i]);
    }
}

template <typename T>
__global__ void sb_log_weights_big_kernel(int P, const T* __restrict__ conc,
                                          const int64_t* __restrict__ ordering,
                                          T* __restrict__ log_w, T* __restrict__ log_1_v_sum) {
    auto l1v = [&](int q) {
        const int i = (int)ordering[q];
        const double a = (double)conc[2 * i], b = (double)conc[2 * i + 1];
        return digamma(b) - digamma(a + b);
    };
    double mine = 0.0;
    for (int r = threadIdx.x; r < P; r += blockDim.x) {
        const int i = (int)ordering[r];
        const double a = (double)conc[2 * i], b = (double)conc[2 * i + 1];
        double acc = digamma(a) - digamma(a + b);
        for (int q = 0; q < r; ++q) acc += l1v(q);
        log_w[i] = (T)acc;
    }
    if (log_1_v_sum) {
        // (the sticks in order, as the small kernel adds them: thread 0 alone)
        if (threadIdx.x == 0) {
            for (int r = 0; r < P; ++r) mine += l1v(r);
            *log_1_v_sum = (T)mine;
        }
    }
}

template <typename T>
int sb_transform_launch(int P, const void* counts, int64_t* ordering, void* stats, void* stream) {
    BEER_REQUIRE(P >= 1 && counts && ordering && stats);
    if (P <= kSbMax)
        hipLaunchKernelGGL(sb_transform_kernel<T>, dim3(1), dim3(256), 0, as_stream(stream), P,
                           (const T*)counts, ordering, (T*)stats);
    else
        hipLaunchKernelGGL(sb_transform_big_kernel<T>, dim3(1), dim3(1024), 0, as_stream(stream), P,
                           (const T*)counts, ordering, (T*)stats);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

template <typename T>
int sb_log_weights_launch(int P, const void* conc, const int64_t* ordering, void* log_w,
                          void* log_1_v_sum, void* stream) {
    BEER_REQUIRE(P >= 1 && conc && ordering && log_w);
    if (P <= kSbMax)
        hipLaunchKernelGGL(sb_log_weights_kernel<T>, dim3(1), dim3(256), 0, as_stream(stream), P,
                           (const T*)conc, ordering, (T*)log_w, (T*)log_1_v_sum);
    else
        hipLaunchKernelGGL(sb_log_weights_big_kernel<T>, dim3(1), dim3(1024), 0, as_stream(stream),
                           P, (const T*)conc, ordering, (T*)log_w, (T*)log_1_v_sum);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

}  // namespace

extern "C" {

int beer_hip_version(void) { return 100; }

int beer_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int beer_nw_expected_stats(int dtype, int K, int D, const void* mean, const void* scale,
                           const void* W, const void* dof, void* out, void* stream) {
    BEER_DISPATCH(dtype, nw_launch, 0, K, D, mean, scale, W, dof, out, stream);
}
int beer_nw_expected_stats_log_norm(int dtype, int K, int D, const void* mean, const void* scale,
                                    const void* W, const void* dof, void* out, void* log_norm,
                                    void* stream) {
    BEER_REQUIRE(log_norm != nullptr || K == 0);
    BEER_DISPATCH(dtype, nw_launch, 0, K, D, mean, scale, W, dof, out, stream, log_norm);
}
int beer_nw_log_norm(int dtype, int K, int D, const void* mean, const void* scale,
                     const void* W, const void* dof, void* out, void* stream) {
    BEER_DISPATCH(dtype, nw_launch, 1, K, D, mean, scale, W, dof, out, stream);
}
int beer_nw_natural(int dtype, int K, int D, const void* mean, const void* scale,
                    const void* W, const void* dof, void* out, void* stream) {
    BEER_DISPATCH(dtype, nw_launch, 2, K, D, mean, scale, W, dof, out, stream);
}
int beer_nw_from_natural(int dtype, int K, int D, const void* eta, void* mean, void* scale,
                         void* W, void* dof, void* stream) {
    BEER_DISPATCH(dtype, nw_from_natural_launch, K, D, eta, mean, scale, W, dof, stream);
}

int beer_nw_update(int dtype, int K, int D, const void* eta, void* mean, void* scale, void* W,
                   void* dof, void* exp_stats, void* log_norm, void* moments, void* stream) {
    BEER_DISPATCH(dtype, nw_update_launch, K, D, eta, mean, scale, W, dof, exp_stats, log_norm,
                  moments, stream);
}

#define NG_ENTRY(NAME, ISO, WHICH)                                                        \
    int NAME(int dtype, int K, int D, const void* mean, const void* scale,                \
             const void* shape, const void* rates, void* out, void* stream) {             \
        if (dtype == BEER_F32)                                                            \
            return ng_launch<float, ISO>(WHICH, K, D, mean, scale, shape, rates, out, stream); \
        if (dtype == BEER_F64)                                                            \
            return ng_launch<double, ISO>(WHICH, K, D, mean, scale, shape, rates, out, stream); \
        return BEER_EINVAL;                                                               \
    }
NG_ENTRY(beer_ng_expected_stats, false, 0)
NG_ENTRY(beer_ng_log_norm, false, 1)
NG_ENTRY(beer_ng_natural, false, 2)
NG_ENTRY(beer_ing_expected_stats, true, 0)
NG_ENTRY(beer_ing_log_norm, true, 1)
NG_ENTRY(beer_ing_natural, true, 2)
#undef NG_ENTRY

int beer_ng_from_natural(int dtype, int K, int D, const void* eta, void* mean, void* scale,
                         void* shape, void* rates, void* stream) {
    if (dtype == BEER_F32) return ng_from_natural_launch<float, false>(K, D, eta, mean, scale, shape, rates, stream);
    if (dtype == BEER_F64) return ng_from_natural_launch<double, false>(K, D, eta, mean, scale, shape, rates, stream);
    return BEER_EINVAL;
}
int beer_ing_from_natural(int dtype, int K, int D, const void* eta, void* mean, void* scale,
                          void* shape, void* rate, void* stream) {
    if (dtype == BEER_F32) return ng_from_natural_launch<float, true>(K, D, eta, mean, scale, shape, rate, stream);
    if (dtype == BEER_F64) return ng_from_natural_launch<double, true>(K, D, eta, mean, scale, shape, rate, stream);
    return BEER_EINVAL;
}

int beer_dirichlet_expected_stats(int dtype, int S, int G, const void* conc, void* out, void* stream) {
    BEER_DISPATCH(dtype, dirichlet_launch, 0, S, G, conc, out, stream);
}
int beer_dirichlet_natural(int dtype, int S, int G, const void* conc, void* out, void* stream) {
    BEER_DISPATCH(dtype, dirichlet_launch, 1, S, G, conc, out, stream);
}
int beer_dirichlet_from_natural(int dtype, int S, int G, const void* eta, void* conc, void* stream) {
    BEER_DISPATCH(dtype, dirichlet_launch, 2, S, G, eta, conc, stream);
}
int beer_dirichlet_log_weights(int dtype, int S, int G, const void* conc, void* out, void* stream) {
    BEER_DISPATCH(dtype, dirichlet_launch, 3, S, G, conc, out, stream);
}
int beer_dirichlet_log_norm(int dtype, int S, int G, const void* conc, void* out, void* stream) {
    BEER_DISPATCH(dtype, dirichlet_launch, 4, S, G, conc, out, stream);
}

int beer_sb_transform_stats(int dtype, int P, const void* counts, int64_t* ordering, void* stats,
                            void* stream) {
    BEER_DISPATCH(dtype, sb_transform_launch, P, counts, ordering, stats, stream);
}
int beer_sb_log_weights(int dtype, int P, const void* conc, const int64_t* ordering, void* log_w,
                        void* log_1_v_sum, void* stream) {
    BEER_DISPATCH(dtype, sb_log_weights_launch, P, conc, ordering, log_w, log_1_v_sum, stream);
}

int beer_gamma_expected_stats(int dtype, int n, const void* shape, const void* rate, void* out, void* stream) {
    BEER_DISPATCH(dtype, gamma_launch, 0, n, shape, rate, out, nullptr, stream);
}
int beer_gamma_natural(int dtype, int n, const void* shape, const void* rate, void* out, void* stream) {
    BEER_DISPATCH(dtype, gamma_launch, 1, n, shape, rate, out, nullptr, stream);
}
int beer_gamma_from_natural(int dtype, int n, const void* eta, void* shape, void* rate, void* stream) {
    BEER_DISPATCH(dtype, gamma_launch, 2, n, eta, nullptr, shape, rate, stream);
}
int beer_gamma_log_norm(int dtype, int n, const void* shape, const void* rate, void* out, void* stream) {
    BEER_DISPATCH(dtype, gamma_launch, 3, n, shape, rate, out, nullptr, stream);
}

int beer_kl_div(int dtype, int K, int Q, const void* es, const void* eq, const void* ep,
                const void* lq, const void* lp, void* out, void* stream) {
    BEER_DISPATCH(dtype, kl_launch, K, Q, es, eq, ep, lq, lp, out, stream);
}

int beer_natural_grad_step(int dtype, int64_t n, const void* eta_prior, const void* eta_post,
                           const void* stats, double lrate, void* out, void* stream) {
    BEER_DISPATCH(dtype, nat_grad_launch, n, eta_prior, eta_post, stats, lrate, out, stream);
}

int beer_suffstats_expand(int dtype, int cov, int64_t T, int D, const void* X, void* out,
                          void* stream) {
    BEER_DISPATCH(dtype, suffstats_launch, cov, T, D, X, out, stream);
}

}  // extern "C"
