// Shared device helpers for the beer_amd HIP kernels (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "beer_hip.h"

#define BEER_WAVE 64

// Launch check: kernels never synchronise; a launch failure is reported as
// -(hipError_t).
#define BEER_LAUNCH_CHECK()                           \
    do {                                              \
        hipError_t e__ = hipGetLastError();           \
        if (e__ != hipSuccess) return -(int)e__;      \
    } while (0)

#define BEER_REQUIRE(cond) \
    do {                   \
        if (!(cond)) return BEER_EINVAL; \
    } while (0)

// dtype dispatch: FN is a template<typename T> int fn(args...).
#define BEER_DISPATCH(dtype, FN, ...)                         \
    do {                                                      \
        if ((dtype) == BEER_F32) return FN<float>(__VA_ARGS__);  \
        if ((dtype) == BEER_F64) return FN<double>(__VA_ARGS__); \
        return BEER_EINVAL;                                   \
    } while (0)

static inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }

namespace beer {

// current value of a BEER_OPT_* tuning option (util.hip)
int option(int key);
constexpr int kAxMaxFramesDefault = 4096;   // BEER_OPT_AX_MAXFRAMES (estep_bf16.hip: accx_kernel)

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of the KERNEL, not of a
// launch: every launch site sets it to the same value (a CU's whole LDS), so host
// threads that share a kernel cannot lower each other's limit between the attribute
// call and the launch.
constexpr int kMaxDynLds = 160 * 1024;

constexpr double kLog2Pi = 1.8378770664093453;
constexpr double kLog2 = 0.6931471805599453;
constexpr double kLogPi = 1.1447298858494002;

__device__ __forceinline__ double neg_inf() { return -__builtin_huge_val(); }

// psi(x), x > 0: upward recurrence to x >= 10 then the asymptotic series
// (error < 1e-15 relative for the arguments the model produces).
__device__ inline double digamma(double x) {
    double r = 0.0;
    while (x < 10.0) {
        r -= 1.0 / x;
        x += 1.0;
    }
    const double f = 1.0 / (x * x);
    const double t = f * (-1.0 / 12.0 + f * (1.0 / 120.0 + f * (-1.0 / 252.0 +
                     f * (1.0 / 240.0 + f * (-1.0 / 132.0 + f * (691.0 / 32760.0 +
                     f * (-1.0 / 12.0)))))));
    return r + log(x) - 0.5 / x + t;
}

// Wave (64 lanes) all-reductions without the LDS crossbar (__shfl_xor is a
// ds_bpermute per 32-bit word and step: 12 dependent LDS round trips for one fp64
// reduction, which sat on the critical path of every forward-backward step):
// inside a row of 16 lanes by DPP (xor 1, xor 2, half-row mirror, row mirror),
// between the 4 rows by gfx950's v_permlane16_swap / v_permlane32_swap.
namespace wave_detail {
template <int CTRL>
__device__ __forceinline__ unsigned dpp_word(unsigned w) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)w, CTRL, 0xf, 0xf, true);
}
template <typename T> struct Words;
template <> struct Words<float> {
    static constexpr int N = 1;
    static __device__ __forceinline__ void split(float v, unsigned (&w)[2]) {
        w[0] = __builtin_bit_cast(unsigned, v);
    }
    static __device__ __forceinline__ float join(const unsigned (&w)[2]) {
        return __builtin_bit_cast(float, w[0]);
    }
};
template <> struct Words<int> {
    static constexpr int N = 1;
    static __device__ __forceinline__ void split(int v, unsigned (&w)[2]) { w[0] = (unsigned)v; }
    static __device__ __forceinline__ int join(const unsigned (&w)[2]) { return (int)w[0]; }
};
template <> struct Words<double> {
    static constexpr int N = 2;
    static __device__ __forceinline__ void split(double v, unsigned (&w)[2]) {
        const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
        w[0] = (unsigned)b;
        w[1] = (unsigned)(b >> 32);
    }
    static __device__ __forceinline__ double join(const unsigned (&w)[2]) {
        return __builtin_bit_cast(double, ((unsigned long long)w[1] << 32) | w[0]);
    }
};
template <typename T, int CTRL>
__device__ __forceinline__ T dpp(T v) {
    unsigned w[2];
    Words<T>::split(v, w);
#pragma unroll
    for (int i = 0; i < Words<T>::N; ++i) w[i] = dpp_word<CTRL>(w[i]);
    return Words<T>::join(w);
}
// the two operands of the next combining step across rows: with both inputs equal
// to v, the swap leaves a = (rows 0,0,2,2), b = (rows 1,1,3,3) [16] resp.
// a = lower half everywhere, b = upper half everywhere [32]
template <typename T, bool HALVES>
__device__ __forceinline__ void cross(T v, T& a, T& b) {
    unsigned w[2], wa[2], wb[2];
    Words<T>::split(v, w);
#pragma unroll
    for (int i = 0; i < Words<T>::N; ++i) {
        if (HALVES) {
            const auto r = __builtin_amdgcn_permlane32_swap(w[i], w[i], false, false);
            wa[i] = r[0]; wb[i] = r[1];
        } else {
            const auto r = __builtin_amdgcn_permlane16_swap(w[i], w[i], false, false);
            wa[i] = r[0]; wb[i] = r[1];
        }
    }
    a = Words<T>::join(wa);
    b = Words<T>::join(wb);
}
template <typename T, typename Op>
__device__ __forceinline__ T allreduce(T v, Op op) {
    v = op(v, dpp<T, 0xB1>(v));          // quad_perm [1,0,3,2]
    v = op(v, dpp<T, 0x4E>(v));          // quad_perm [2,3,0,1]
    v = op(v, dpp<T, 0x141>(v));         // row_half_mirror
    v = op(v, dpp<T, 0x140>(v));         // row_mirror
    T a, b;
    cross<T, false>(v, a, b);
    v = op(a, b);
    cross<T, true>(v, a, b);
    return op(a, b);
}
}  // namespace wave_detail

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
    return wave_detail::allreduce(v, [](T x, T y) { return x + y; });
}
template <typename T>
__device__ __forceinline__ T wave_max(T v) {
    return wave_detail::allreduce(v, [](T x, T y) { return y > x ? y : x; });
}

// Block reductions through LDS scratch of >= blockDim/64 elements.  All
// threads get the result.  `scratch` is reused: callers must not alias it.
template <typename T>
__device__ inline T block_sum(T v, T* scratch) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    T r = 0;
    for (int i = 0; i < nw; ++i) r += scratch[i];
    return r;
}
template <typename T>
__device__ inline T block_max(T v, T* scratch) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    T r = scratch[0];
    for (int i = 1; i < nw; ++i) r = scratch[i] > r ? scratch[i] : r;
    return r;
}

__host__ __device__ inline int stats_dim(int cov, int D) {
    return cov == BEER_FULL ? D * D + D + 2 : (cov == BEER_DIAG ? 2 * D + 2 : D + 3);
}

}  // namespace beer

// ---------------------------------------------------------------------------
// In-LDS SPD inverse (one workgroup per matrix, A is D x D row-major fp64 in LDS).
// ---------------------------------------------------------------------------
// In-place inverse of the SPD matrix A by Gauss-Jordan elimination without
// pivoting (backward stable for SPD matrices); returns log|A| = sum of the log
// pivots (all threads).  Two barriers per column and D*D / nt entry updates per
// thread and column -- against a Cholesky factorisation plus a triangular
// inverse whose substitution runs D*D / 2 dependent LDS reads deep in ONE thread
// per column (76 -> ~20 us per launch at D = 40: these kernels are pure latency,
// one workgroup per matrix).  `cr` = LDS scratch of 2 D doubles.  With
// `want_inverse` false only the trailing (Schur) updates are made: log|A| alone.
__device__ inline double spd_inverse(double* A, int D, double* cr, bool want_inverse) {
    const int tid = threadIdx.x, nt = blockDim.x;
    double* col = cr;
    double* row = cr + D;
    const int i0 = tid / D, k0 = tid - i0 * D, di = nt / D, dk = nt - di * D;
    double mant = 1.0;                      // log|A| = log(mant) + expo * log 2
    int expo = 0;
    for (int j = 0; j < D; ++j) {
        __syncthreads();
        const double p = A[j * D + j], ip = 1.0 / p;
        int e;
        mant *= frexp(p, &e);
        expo += e;
        if ((j & 31) == 31) { mant = frexp(mant, &e); expo += e; }
        // log|A| alone: the lower triangle is enough (column j serves as row j)
        for (int i = tid; i < D; i += nt) {
            col[i] = A[i * D + j];
            row[i] = (want_inverse ? A[j * D + i] : A[i * D + j]) * ip;
        }
        __syncthreads();
        int i = i0, k = k0;
        for (int idx = tid; idx < D * D; idx += nt) {
            if (want_inverse) {
                double v;
                if (i == j) v = (k == j) ? ip : row[k];
                else if (k == j) v = -col[i] * ip;
                else v = A[idx] - col[i] * row[k];
                A[idx] = v;
            } else if (k > j && k <= i) {
                A[idx] -= col[i] * row[k];
            }
            i += di;
            k += dk;
            if (k >= D) { k -= D; ++i; }
        }
    }
    __syncthreads();
    return log(mant) + (double)expo * 0.69314718055994530942;
}
