// Do the MFMAs of one wave overlap the VALU / transcendental work of ANOTHER wave on the same
// SIMD (gfx950)?  One workgroup per CU, 8 waves = two per SIMD (waves w and w + 4 share a SIMD).
// Modes:  0  all 8 waves: MFMA stream only (NM MFMAs on 8 independent accumulators)
//         1  all 8 waves: VALU stream only (NV dependent-free v_fma / v_exp mix)
//         2  waves 0-3 MFMA stream, waves 4-7 VALU stream (the cross-wave overlap case)
//         3  all 8 waves: MFMA stream then VALU stream (phases, as the E-step kernels do)
//         4  all 8 waves: ONE stream, a VALU slice behind every MFMA (single-wave interleave)
//         5  waves 0-3 only: MFMA stream (one wave per SIMD: the pipe's own rate)
//         6  waves 0-3 only: interleaved stream of mode 4
// Prints cycles per workgroup (s_memtime) for each mode; every TRANS-th filler is a v_add + v_exp pair (TRANS=0: none).
// Build: hipcc --offload-arch=gfx950 -O2 tools/probes/coissue.hip -o gpurun_out/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NM = 2048;          // MFMAs per wave
#ifndef FILL
#define FILL 2                    // VALU instructions per MFMA in the VALU stream / the interleave
#endif
#ifndef TRANS
#define TRANS 5                   // every TRANS-th filler is a v_exp_f32 (0: none)
#endif

// (values stay bounded: v <- v/2 + c converges to 2 c; the exponentials take 2^(v' - 1) of a
// neighbour, in [1/2, 2] -- NaNs and infinities made v_exp_f32 hundreds of cycles slow in
// the first version of this probe)
template <int N>
__device__ __forceinline__ void valu_slice(float (&v)[8], int n) {
    const float half = 0.5f, c = 0.37f, m1 = -1.0f;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const int idx = (n * N + j) & 7;
        if (TRANS && ((n * N + j) % TRANS == TRANS - 1)) {
            float t;
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(t) : "v"(v[(idx + 1) & 7]), "v"(m1));
            asm volatile("v_exp_f32 %0, %1" : "=v"(v[idx]) : "v"(t));
        } else {
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[idx]) : "v"(half), "v"(c));
        }
    }
}

__global__ __launch_bounds__(512, 2) void k(int mode, float* out, long long* cycles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bf8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(1.0f + lane * 0.001f); b[e] = (__bf16)(0.5f); }
    f32x4 acc[8];
    for (int c = 0; c < 8; ++c) acc[c] = f32x4{0, 0, 0, 0};
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = 0.001f * (lane + j);
    const bool second = wave >= 4;
    if ((mode == 5 || mode == 6) && second) return;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    const bool do_m = mode == 0 || mode == 3 || mode == 5 || (mode == 2 && !second);
    const bool do_v = mode == 1 || mode == 3 || (mode == 2 && second);
    if (mode == 4 || mode == 6) {
#pragma unroll 16
        for (int n = 0; n < NM; ++n) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[n & 7]) : "v"(a), "v"(b));
            valu_slice<FILL>(v, n);
        }
    } else {
        if (do_m) {
#pragma unroll 16
            for (int n = 0; n < NM; ++n)
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[n & 7]) : "v"(a), "v"(b));
        }
        if (do_v) {
#pragma unroll 16
            for (int n = 0; n < NM; ++n) valu_slice<FILL>(v, n);
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    __syncthreads();
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int c = 0; c < 8; ++c) s += acc[c][0] + acc[c][3];
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
    float* out; long long* cyc;
    const int nb = 256;
    hipMalloc(&out, nb * 512 * 4); hipMalloc(&cyc, nb * 8);
    long long h[nb];
    const char* names[] = {"8 waves: MFMA only", "8 waves: VALU only", "4 waves MFMA + 4 waves VALU (pairs per SIMD)",
                           "8 waves: MFMA phase then VALU phase", "8 waves: interleaved in one stream",
                           "4 waves (1 per SIMD): MFMA only", "4 waves (1 per SIMD): interleaved"};
    printf("NM %d MFMAs (16 cycles each alone = %d), FILL %d VALU per MFMA, TRANS %d\n", NM, NM * 16, FILL, TRANS);
    for (int mode = 0; mode < 7; ++mode) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(nb), dim3(512), 0, 0, mode, out, cyc);
        hipDeviceSynchronize();
        hipMemcpy(h, cyc, nb * 8, hipMemcpyDeviceToHost);
        double m = 0; for (int i = 0; i < nb; ++i) m += h[i];
        printf("mode %d  %-48s %9.0f clock ticks per workgroup\n", mode, names[mode], m / nb);
    }
    return 0;
}
