// MFMA (matrix-core) fast path for the full-covariance E-step on gfx950.
// See estep_mfma.hip for the design; estep.hip routes to it when
// `supported(D, K)` holds.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace beer_mfma {

bool supported(int D, int K);

int llh_full_f32(int64_t T, int D, int K, const float* X, const float* expT, const float* logw,
                 float* w_out, hipStream_t s);
int llh_full_f64(int64_t T, int D, int K, const double* X, const double* expT,
                 const double* logw, double* w_out, hipStream_t s);
int acc_full_f32(int64_t T, int D, int S, int G, const float* X, const float* comp_resps,
                 const float* state_resps, double* acc, hipStream_t s);
int acc_full_f64(int64_t T, int D, int S, int G, const double* X, const double* comp_resps,
                 const double* state_resps, double* acc, hipStream_t s);

}  // namespace beer_mfma
