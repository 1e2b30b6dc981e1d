// MFMA (matrix-core) fast path of the E-step on gfx950 (all three covariance
// types).  See estep_mfma.hip for the design; estep.hip routes to it when the
// shape is supported and the caller provides the scratch workspace.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace beer_mfma {

// Largest feature dimension of the exact matrix-core paths: float32 96 (the accumulation
// kernels keep two tiles of transposed frames + responsibilities in a CU's 160 KiB of LDS),
// float64 64.  Beyond: the generic kernels of estep.hip.
constexpr int kMaxDimF32 = 96, kMaxDimF64 = 64;
// ... and of the bf16x3 kernels (estep_bf16.hip): 128 -- beyond 112 dimensions the packed
// accumulation keeps ONE tile of transposed frames in LDS instead of two (accx_kernel: sx).
constexpr int kMaxDimX = 128;
inline int max_dim(size_t elem) { return elem == 8 ? kMaxDimF64 : kMaxDimF32; }
bool supported_llh(int D, int S, int G, size_t elem = 4);
bool supported_acc(int D, int K, size_t elem = 4);
bool supported_llh_x(int D, int S, int G);           // the same shapes with D <= kMaxDimX
bool supported_acc_x(int D, int K);
size_t estep_workspace_bytes(size_t elem, int cov, int D, int S, int G);
size_t acc_workspace_bytes(int cov, int D, int K, size_t elem = 4);

// fp32 models on the bf16 matrix pipes (estep_bf16.hip): every fp32 operand is held
// exactly as three bf16 pieces, the six leading partial products of every
// multiplication, fp32 accumulation.
size_t estepx_workspace_bytes(int cov, int D, int S, int G);
// shapes that path takes when NO responsibilities are wanted: mixture sets with
// any number of components per state (groups padded to a power of two)
bool supported_llh_split(int D, int S, int G);
// `packed`: S = 1, or a set that is supported_llh_packed_sets; `resps`
// (packed_resps_bytes) then receives the three-plane bf16 image the accumulation
// kernel consumes (estep_tiles.h: softmax_epilogue<PACKED>).
int estep_bf16x3(int cov, int64_t T, int D, int S, int G, const float* X, const float* expT,
                 const float* logw, float* resps, float* log_norm, double* llh_sum, void* ws,
                 size_t ws_bytes, hipStream_t s, bool packed = false,
                 const void* frame_image = nullptr);

int unpack_resps(int64_t T, int K, const void* packed, float* resps, hipStream_t s);
// comp_resps [T, S*G] (x state_resps [T, S], nullable) -> packed tiles
int pack_resps(int64_t T, int D, int S, int G, const float* X, const float* R, const float* SR,
               void* packed, hipStream_t s);
// The responsibilities as estep_bf16x3(..., packed = true) wrote them
// (packed_resps_bytes); the workspace also holds the transposed frames, hence T.
size_t packed_resps_bytes(int64_t T, int D, int K);
size_t accx_workspace_bytes(int cov, int64_t T, int D, int K);
int acc_bf16x3_packed(int cov, int64_t T, int D, int K, const float* X, const void* Rimg,
                      double* acc, void* ws, size_t ws_bytes, hipStream_t s, int S = 1, int G = 0,
                      const float* SR = nullptr);
// Mixture sets on the packed hand-over (full covariance): the E-step leaves the
// responsibilities within each state's mixture as packed tiles (estep_bf16x3 with
// packed = true, S > 1), the accumulation multiplies the state posteriors SR [T, S]
// in while a tile sits in LDS (workspace: accxs_workspace_bytes).
bool supported_llh_packed_sets(int cov, int D, int S, int G);
bool supported_acc_sets(int cov, int D, int S, int G);
size_t accxs_workspace_bytes(int cov, int64_t T, int D, int S, int G);

// Mixture sets with diagonal / isotropic Gaussians: accumulation that recomputes the
// component responsibilities from the frames and the per-state log-normalisers
// [T, S] of the E-step (times the state responsibilities sr [T, S], nullable)
// instead of reading them from memory (estep_bf16.hip: accf_kernel).
bool supported_accf(int cov, int D, int S, int G);
size_t accf_workspace_bytes(int cov, int D, int S, int G);
int acc_fused_bf16x3(int cov, int64_t T, int D, int S, int G, const float* X, const float* expT,
                     const float* logw, const float* log_norm, const float* sr, const void* image,
                     double* acc, void* ws, size_t ws_bytes, hipStream_t s);
// The operands of that kernel that depend on the frames only, built once per block of
// frames (estep_bf16.hip: frame_image_kernel); `image` above, nullable.
size_t frame_image_bytes(int cov, int64_t T, int D);
int frame_image(int cov, int64_t T, int D, const float* X, void* image, hipStream_t s);

// Diagonal / isotropic Gaussians, float32 weights [T, K] in memory (acc_diag.hip: accd_kernel):
// acc += W^T phi(X) in the bf16x3 arithmetic.  Workspace (optional): the partial sums of every
// chain of frames, added up by a second small kernel; without it fp64 atomics.
bool supported_acc_diag(int cov, int64_t T, int D, int K);
size_t acc_diag_workspace_bytes(int cov, int64_t T, int D, int K);
int acc_diag_bf16x3(int cov, int64_t T, int D, int K, const float* X, const float* W, double* acc,
                    void* ws, size_t ws_bytes, hipStream_t s);

int estep_f32(int cov, int64_t T, int D, int S, int G, const float* X, const float* expT,
              const float* logw, float* resps, float* log_norm, double* llh_sum, void* ws,
              size_t ws_bytes, hipStream_t s);
int estep_f64(int cov, int64_t T, int D, int S, int G, const double* X, const double* expT,
              const double* logw, double* resps, double* log_norm, double* llh_sum, void* ws,
              size_t ws_bytes, hipStream_t s);
int acc_f32(int cov, int64_t T, int D, int S, int G, const float* X, const float* R,
            const float* SR, double* acc, void* ws, size_t ws_bytes, hipStream_t s);
int acc_f64(int cov, int64_t T, int D, int S, int G, const double* X, const double* R,
            const double* SR, double* acc, void* ws, size_t ws_bytes, hipStream_t s);

}  // namespace beer_mfma
