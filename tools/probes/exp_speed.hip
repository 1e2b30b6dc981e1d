// Does v_exp_f32's speed depend on its argument?  (A first version of coissue.hip let its values
// run into NaN / infinity and measured hundreds of cycles per exponential.)  8 waves per
// workgroup, one workgroup per CU, 4096 v_exp_f32 per wave on a constant argument.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probes/exp_speed.hip -o build_ab/probes/exp_speed
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__global__ __launch_bounds__(512) void k(float arg, float* out, long long* cycles) {
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = arg;
    float s = 0.f;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 8
    for (int n = 0; n < 4096; ++n) {
        float r;
        asm volatile("v_exp_f32 %0, %1" : "=v"(r) : "v"(v[n & 7]));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(s) : "v"(r));
    }
    __syncthreads();
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}
int main() {
    float* out; long long* cyc;
    const int nb = 256;
    hipMalloc(&out, nb * 512 * 4); hipMalloc(&cyc, nb * 8);
    long long h[nb];
    const float args[] = {-1.f, -100.f, -126.5f, -130.f, -140.f, -148.f, -151.f, -200.f, -1e30f, -INFINITY, NAN, 3.f, 200.f};
    for (float a : args) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(nb), dim3(512), 0, 0, a, out, cyc);
        hipDeviceSynchronize();
        hipMemcpy(h, cyc, nb * 8, hipMemcpyDeviceToHost);
        double m = 0; for (int i = 0; i < nb; ++i) m += h[i];
        printf("v_exp_f32(%12g) + v_add: %8.1f cycles per pair and wave (8 waves per CU)\n", a, m / nb / 4096);
    }
    return 0;
}
