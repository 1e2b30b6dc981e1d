// Issue rate of the bf16 MFMA shapes on gfx950, one wave per SIMD, 8 independent accumulators:
// v_mfma_f32_16x16x32_bf16 (8 bf16 per lane and operand) against the older
// v_mfma_f32_16x16x16_bf16 (4 per lane) and v_mfma_f32_32x32x16_bf16 / 32x32x8.  Does a half-depth
// k-step cost half?  (the logits of a diagonal mixture have 80 statistic columns: 2.5 k-steps of 32)
// Build: hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_k16.hip -o gpurun_out/mfma_k16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int NM = 4096;

__global__ __launch_bounds__(256) void k(int mode, float* out, long long* cycles) {
    const int lane = threadIdx.x & 63;
    bf8 a8, b8; bf4 a4, b4;
    for (int e = 0; e < 8; ++e) { a8[e] = (__bf16)(1.0f + lane * 0.001f); b8[e] = (__bf16)0.5f; }
    for (int e = 0; e < 4; ++e) { a4[e] = a8[e]; b4[e] = b8[e]; }
    f32x4 acc[8];
    f32x16 big[2];
    for (int c = 0; c < 8; ++c) acc[c] = f32x4{0, 0, 0, 0};
    for (int c = 0; c < 2; ++c) for (int e = 0; e < 16; ++e) big[c][e] = 0.f;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
#define M32(c) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(a8), "v"(b8))
#define M16(c) asm volatile("v_mfma_f32_16x16x16_bf16 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(a4), "v"(b4))
#define B16(c) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(big[c]) : "v"(a8), "v"(b8))
#define B8(c) asm volatile("v_mfma_f32_32x32x8_bf16 %0, %1, %2, %0" : "+v"(big[c]) : "v"(a4), "v"(b4))
    if (mode == 0) {
        for (int n = 0; n < NM / 8; ++n) { M32(0); M32(1); M32(2); M32(3); M32(4); M32(5); M32(6); M32(7); }
    } else if (mode == 1) {
        for (int n = 0; n < NM / 8; ++n) { M16(0); M16(1); M16(2); M16(3); M16(4); M16(5); M16(6); M16(7); }
    } else if (mode == 2) {
        for (int n = 0; n < NM / 8; ++n) { B16(0); B16(1); B16(0); B16(1); B16(0); B16(1); B16(0); B16(1); }
    } else if (mode == 3) {
        for (int n = 0; n < NM / 8; ++n) { B8(0); B8(1); B8(0); B8(1); B8(0); B8(1); B8(0); B8(1); }
    } else {
        // two full-depth steps and one half-depth step, as a row of 80 columns would take
        for (int n = 0; n < NM / 12; ++n) {
            M32(0); M32(1); M16(2); M32(3); M32(4); M16(5); M32(6); M32(7); M16(0);
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    __syncthreads();
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int c = 0; c < 8; ++c) s += acc[c][0] + acc[c][3];
    for (int c = 0; c < 2; ++c) s += big[c][0] + big[c][15];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
    float* out; long long* cyc;
    const int nb = 256;
    hipMalloc(&out, nb * 256 * 4); hipMalloc(&cyc, nb * 8);
    long long h[nb];
    const char* names[] = {"16x16x32_bf16", "16x16x16_bf16", "32x32x16_bf16", "32x32x8_bf16",
                           "2 x 16x16x32 + 1 x 16x16x16 (per 3)"};
    const int count[] = {NM, NM, NM, NM, NM / 12 * 9};
    for (int warm = 0; warm < 20; ++warm) hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, 1, out, cyc);   // clocks up
    for (int mode = 0; mode < 5; ++mode) {
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, mode, out, cyc);
        hipDeviceSynchronize();
        hipMemcpy(h, cyc, nb * 8, hipMemcpyDeviceToHost);
        double m = 0; for (int i = 0; i < nb; ++i) m += h[i];
        printf("%-40s %8.2f clock ticks per MFMA (one wave per SIMD)\n", names[mode], m / nb / count[mode]);
    }
    return 0;
}
