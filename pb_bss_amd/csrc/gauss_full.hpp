// Host-callable launchers of the full-covariance Gaussian kernels (gauss_full.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include "pbbss.h"
#include "embed.hpp"  // PartialReduce

namespace pbbss {
constexpr int kGaussFullMaxE = 63;  // augmented vector [y - c; 1] in at most four 16-blocks

// scratch of launch_gauss_full_fit (wave partials of the Gram tiles), in doubles
size_t gauss_full_partial_doubles(int64_t B, int64_t N, int E, int K);

// GaussianTrainer._fit, covariance_type='full' (gaussian.py:152-193): y (B,N,E) row-major,
// weights (B,K,N) -> out_mean (B,K,E), out_cov (B,K,E,E); with out_mq / out_offset non-null also
// the factorisation the log-pdf needs (Mq = X X^T, X = L^-1; offset = -E/2 ln 2pi - sum ln L_dd)
// and PBBSS_ST_NOT_POSDEF or-ed into *out_status where a covariance is not positive definite.
// sal (B,N) or null multiplies the weights (affiliation * saliency); out_s0 (B,K) or null
// receives the weight sums (for the mixture weights).
int launch_gauss_full_fit(const void* y, int y_is_f64, int64_t B, int64_t N, int E, int K,
                          const double* weights, const double* sal, double* part,
                          double* out_mean, double* out_cov, double* out_mq, double* out_offset,
                          double* out_s0, int32_t* out_status, hipStream_t s,
                          // bins sharded over ranks (B = 1): `shift` (E) is the common centre c of
                          // the augmented vectors (null: row 0 of y), `reduce` sums the Gram tiles
                          // over the ranks between the reduction and the finalize kernel
                          const double* shift = nullptr, const PartialReduce* reduce = nullptr);
// mixture weights from the weight sums: mode 0 L1-normalised over the classes, 1 uniform
int launch_gauss_full_weights(const double* s0, int64_t B, int K, int mode, double* out_weight,
                              hipStream_t s);
// the same factorisation for given covariances (BK matrices)
int launch_gauss_full_factor(const double* cov, int64_t BK, int E, double* out_mq,
                             double* out_offset, int32_t* out_status, hipStream_t s);
// Gaussian.log_pdf (gaussian.py:35-56) and, with weight (B,K), the posteriors of the mixture
int launch_gauss_full_logpdf(const void* y, int y_is_f64, int64_t B, int64_t N, int E, int K,
                             const double* mean, const double* mq, const double* offset,
                             const double* weight, double* out_lp, double* out_aff, hipStream_t s);
}  // namespace pbbss
