// Placeholder until the MFMA path lands: nothing is supported, the generic
// kernels of estep.hip serve every shape.
#include "estep_mfma.h"
#include "beer_hip.h"

namespace beer_mfma {
bool supported(int, int) { return false; }
int llh_full_f32(int64_t, int, int, const float*, const float*, const float*, float*, hipStream_t) { return BEER_EINVAL; }
int llh_full_f64(int64_t, int, int, const double*, const double*, const double*, double*, hipStream_t) { return BEER_EINVAL; }
int acc_full_f32(int64_t, int, int, int, const float*, const float*, const float*, double*, hipStream_t) { return BEER_EINVAL; }
int acc_full_f64(int64_t, int, int, int, const double*, const double*, const double*, double*, hipStream_t) { return BEER_EINVAL; }
}  // namespace beer_mfma
