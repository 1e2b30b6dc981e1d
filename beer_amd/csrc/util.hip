// Small device utilities of the host layer.

#include "common.h"

namespace {

__global__ void copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16;
         i += (size_t)gridDim.x * blockDim.x)
        dst[i] = src[i];
}

}  // namespace

extern "C" {

int beer_copy_pinned(void* dst_device, const void* src_pinned_host, size_t nbytes, void* stream) {
    if (nbytes == 0) return BEER_OK;
    if (!dst_device || !src_pinned_host || (nbytes & 15) ||
        ((uintptr_t)dst_device & 15) || ((uintptr_t)src_pinned_host & 15))
        return BEER_EINVAL;
    const size_t n16 = nbytes / 16;
    size_t blocks = (n16 + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(copy_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream),
                       (const uint4*)src_pinned_host, (uint4*)dst_device, n16);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

}  // extern "C"
