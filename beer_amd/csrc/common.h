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

// Wave (64 lanes) reductions.
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        T w = __shfl_xor(v, o, 64);
        v = w > v ? w : v;
    }
    return v;
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
