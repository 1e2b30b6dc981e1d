// What does v_mfma_f32_16x16x32_bf16 do inside?  D = C + sum_k A[i][k] B[k][j] for random
// bf16 A, B and a float32 C of chosen magnitude, against the exactly rounded result
// (double arithmetic is exact here: 32 products of 16 significant bits).  Reports the mean
// (bias) and the RMS of the error in units of ulp(C), for products of decreasing size.
// Build: hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_round.hip -o gpurun_out/mfma_round
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__global__ void k(const unsigned short* A, const unsigned short* B, const float* C, float* D, int chain) {
    // A [16][32] row-major bf16 bits, B [32][16], C/D [16][16]; `chain` MFMAs in a row (same A, B)
    const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
    unsigned short a[8], b[8];
    for (int e = 0; e < 8; ++e) { a[e] = A[i * 32 + 8 * g + e]; b[e] = B[(8 * g + e) * 16 + i]; }
    bf8 av, bv;
    memcpy(&av, a, 16); memcpy(&bv, b, 16);
    f32x4 c;
    for (int r = 0; r < 4; ++r) c[r] = C[(4 * g + r) * 16 + i];
    for (int n = 0; n < chain; ++n) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + i] = c[r];
}
static unsigned short bf(float v) { unsigned u; memcpy(&u, &v, 4); u += 0x7fff + ((u >> 16) & 1); return u >> 16; }
static float f(unsigned short b) { unsigned u = (unsigned)b << 16; float v; memcpy(&v, &u, 4); return v; }
int main() {
    unsigned short *dA, *dB; float *dC, *dD;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dC, 1024); hipMalloc(&dD, 1024);
    srand(1);
    for (int mode = 0; mode < 2; ++mode)           // 0: random signs, 1: all products positive
    for (int cexp = 7; cexp >= -20; cexp -= 27)     // C ~ 2^7 (large accumulator) or 2^-20 (tiny)
    for (int pexp = 4; pexp >= -28; pexp -= 4) {    // products ~ 2^pexp
        double sum = 0, sum2 = 0, sumr = 0; int n = 0;
        for (int trial = 0; trial < 64; ++trial) {
            std::vector<unsigned short> A(512), B(512); std::vector<float> C(256), D(256);
            for (int x = 0; x < 512; ++x) {
                float va = (1.f + rand() / (float)RAND_MAX) * (mode || rand() % 2 ? 1.f : -1.f);
                float vb = ldexpf(1.f + rand() / (float)RAND_MAX, pexp) ;
                A[x] = bf(va); B[x] = bf(vb);
            }
            for (int x = 0; x < 256; ++x) C[x] = ldexpf(1.f + rand() / (float)RAND_MAX, cexp);
            hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
            hipMemcpy(dC, C.data(), 1024, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, 1);
            hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
            for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
                double ex = C[i * 16 + j];
                for (int kk = 0; kk < 32; ++kk) ex += (double)f(A[i * 32 + kk]) * (double)f(B[kk * 16 + j]);
                const float rn = (float)ex;                       // correctly rounded
                int e; frexp(ex, &e); const double ulp = ldexp(1.0, e - 24);
                const double err = ((double)D[i * 16 + j] - ex) / ulp, errr = ((double)rn - ex) / ulp;
                sum += err; sum2 += err * err; sumr += errr * errr; ++n;
            }
        }
        printf("mode %d C~2^%-3d products~2^%-3d: bias %+8.4f ulp  rms %8.4f ulp  (correct rounding: rms %6.4f)\n",
               mode, cexp, pexp, sum / n, sqrt(sum2 / n), sqrt(sumr / n));
    }
    return 0;
}
