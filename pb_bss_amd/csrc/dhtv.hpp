// Host-callable launchers of the permutation-alignment kernels (dhtv.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pbbss.h"

namespace pbbss {
// team: for few utterances several workgroups share one utterance (dhtv_team_kernel);
// team_buf = handle-owned device memory of team_bytes (control words + centroid partials),
// null / too small / team_size <= 1 selects the one-workgroup-per-utterance kernel.
constexpr int kDhtvTeamMax = 32;
int launch_dhtv(const double* mask, int64_t U, int K, int F, int T, const int32_t* plan, int P,
                int optimal, int metric, double* feat, int32_t* mapping, int32_t* status,
                size_t lds_limit, int num_cu, int team_size, void* team_buf, size_t team_bytes,
                int probe, unsigned spin_limit, hipStream_t s);  // probe: frame-slice path, see
                // dhtv_slice_kernel; spin_limit: polls before a team barrier gives up (0: default)
int launch_apply_mapping(const double* mask, const int32_t* mapping, int64_t U, int K, int F,
                         int T, double* out, hipStream_t s);
// pairwise solvers (Oracle / Greedy alignment) and the assignment on given score matrices
int launch_pa_pair(const double* mask, const double* ref, int64_t U, int K, int64_t F, int T,
                   const int64_t* mask_strides, const int64_t* ref_strides, int metric,
                   int optimal, double* scores, int32_t* mapping, int64_t map_F, int64_t map_col0,
                   int32_t* status, hipStream_t s);
int launch_pa_compose(int32_t* mapping, int64_t U, int K, int64_t F, hipStream_t s);
int launch_pa_assign(const double* scores, int64_t N, int K, int optimal, int32_t* mapping,
                     int32_t* status, hipStream_t s);
}  // namespace pbbss
