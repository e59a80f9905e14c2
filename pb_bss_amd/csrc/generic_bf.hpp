// Generic-size (9 <= D <= 32 sensors) beamformer kernels: host-callable launchers
// (generic_bf.hip); arguments as the D <= 8 launchers of beamform.hpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pbbss.h"

namespace pbbss {
int launch_gen_solve(const double* A, const double* Bm, int64_t N, int D, int M, double* x,
                     int32_t* st, size_t lds_limit, hipStream_t s);
int launch_gen_souden(const double* t, const double* nn, int64_t N, int D, double eps, int mode,
                      double* mat, double* num, double* den, int32_t* st, size_t lds_limit,
                      hipStream_t s);
int launch_gen_mvdr(const double* atf, const double* nn, int64_t N, int D, double* w, int32_t* st,
                    size_t lds_limit, hipStream_t s);
int launch_gen_ban(const double* w, const double* nn, int64_t N, int D, double* out, hipStream_t s);
int launch_gen_gev(const double* t, const double* nn, int64_t N, int D, double* w, int32_t* st,
                   size_t lds_limit, hipStream_t s);
}  // namespace pbbss
