// Small device utilities of the host layer.

#include <atomic>

#include "common.h"

namespace beer {
// Tuning options (beer_hip_set_option): process-wide, read at launch time by the host
// code of the kernels they tune.  The defaults are what the parity suite validates.
namespace {
struct OptSpec { int def, lo, hi; };
constexpr OptSpec kOptSpec[BEER_OPT_COUNT] = {
    {kAxMaxFramesDefault, 64, 1 << 20},      // BEER_OPT_AX_MAXFRAMES
    {6, 1, 64},               // BEER_OPT_ACCF_ROUNDS
    {0, 0, 1},                // BEER_OPT_K1_WIDE
    {4, 4, 8},                // BEER_OPT_ACCFI_WAVES (4 or 8)
    {1, 0, 1},                // BEER_OPT_LNFI
    {0, 0, 1},                // BEER_OPT_FB_LOG
    {1, 0, 1},                // BEER_OPT_K1_LDS
};
static_assert(sizeof(kOptSpec) / sizeof(kOptSpec[0]) == BEER_OPT_COUNT, "one OptSpec per option");
// every option starts at its spec'd default: filled from the table, never listed by hand
struct OptTable {
    std::atomic<int> v[BEER_OPT_COUNT];
    OptTable() { for (int i = 0; i < BEER_OPT_COUNT; ++i) v[i].store(kOptSpec[i].def, std::memory_order_relaxed); }
};
OptTable g_tab;
std::atomic<int>* const g_opt = g_tab.v;
}  // namespace
int option(int key) { return g_opt[key].load(std::memory_order_relaxed); }
}  // namespace beer

namespace {

__global__ void copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16;
         i += (size_t)gridDim.x * blockDim.x)
        dst[i] = src[i];
}

// One wave asleep between two readings of the shader clock and of the reference clock.
__global__ void clock_probe_kernel(int64_t* __restrict__ out, int sleeps) {
    const uint64_t t0 = __builtin_readcyclecounter();
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < sleeps; ++i) __builtin_amdgcn_s_sleep(127);
    const uint64_t t1 = __builtin_readcyclecounter();
    const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[0] = (int64_t)(t1 - t0);
        out[1] = (int64_t)(r1 - r0);
    }
}

}  // namespace

extern "C" {

int beer_clock_probe(int64_t* ticks_out, int32_t sleeps, void* stream) {
    if (!ticks_out || sleeps < 1 || sleeps > (1 << 22)) return BEER_EINVAL;
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, as_stream(stream), ticks_out, sleeps);
    BEER_LAUNCH_CHECK();
    return BEER_OK;
}

int beer_hip_set_option(int option, int value) {
    if (option < 0 || option >= BEER_OPT_COUNT) return BEER_EINVAL;
    const auto& sp = beer::kOptSpec[option];
    if (value < sp.lo || value > sp.hi) return BEER_EINVAL;
    if (option == BEER_OPT_ACCFI_WAVES && value != 4 && value != 8) return BEER_EINVAL;
    beer::g_opt[option].store(value, std::memory_order_relaxed);
    return BEER_OK;
}

int beer_hip_get_option(int option) {
    if (option < 0 || option >= BEER_OPT_COUNT) return BEER_EINVAL;
    return beer::option(option);
}

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
