// Generic-size (9 <= D <= 32 sensors) cACGMM kernels: host-callable launchers (generic.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pbbss.h"

namespace pbbss {

constexpr int kGenMaxD = 32;

bool gen_supported(int D, int K);

// a2-a4: posteriors / quadratic form / log-pdf from an eigen-parameterised model
int launch_gen_estep(const void* y, int y_is_c128, int layout, int64_t B, int T, int D, int K,
                     const double* eigvec, const double* eigval, const double* weight, int64_t wb,
                     int64_t wk, int64_t wt, const uint8_t* activity, double eps, double* out_aff,
                     double* out_q, double* out_logpdf, size_t lds_limit, hipStream_t s);

// a6 / a10: weighted covariances.  mode 0: M-step (D * sum_t gamma sal / q y y^H / sum gamma sal,
// observation unit-normalised when layout is TD); mode 1: PSD with the mask normalised by
// max(sum_t mask, 1e-10); mode 2: PSD plain sums (/T without a mask).  out_weight: mixture
// weights of the M-step (a5), out_sum: class sums.
int launch_gen_cov(const void* y, int y_is_c128, int layout, int64_t B, int T, int D, int K,
                   const double* gamma, int64_t gamma_bstride, const double* q,
                   const double* saliency, int mode, int weight_mode, double* out_cov,
                   double* out_weight, double* out_sum, size_t lds_limit, hipStream_t s);

// a7: Hermitian eigendecomposition (ascending, eigenvectors in columns).  covariance_norm < 0:
// plain numpy.linalg.eigh; otherwise the normalisation / floor of from_covariance.
int launch_gen_heev(const double* a, int64_t N, int D, int covariance_norm, double eig_floor,
                    double* out_val, double* out_vec, int32_t* out_status, size_t lds_limit,
                    hipStream_t s);

}  // namespace pbbss
