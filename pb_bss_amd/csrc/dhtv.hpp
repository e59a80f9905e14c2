// Host-callable launchers of the permutation-alignment kernels (dhtv.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pbbss.h"

namespace pbbss {
int launch_dhtv(const double* mask, int64_t U, int K, int F, int T, const int32_t* plan, int P,
                int optimal, double* feat, int32_t* mapping, int32_t* status, size_t lds_limit,
                hipStream_t s);
int launch_apply_mapping(const double* mask, const int32_t* mapping, int64_t U, int K, int F,
                         int T, double* out, hipStream_t s);
}  // namespace pbbss
