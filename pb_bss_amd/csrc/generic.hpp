// Generic-size (9 <= D <= 32 sensors) cACGMM kernels: host-callable launchers (generic.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pbbss.h"

namespace pbbss {

constexpr int kGenMaxD = 32;

bool gen_supported(int D, int K);

// The E-step consumes the model as the "inverse state": B^-1 (complex128 (N, LD, LD) with
// LD = gen_state_ld(D), zero beyond D) and log det B of matrix n = b * K + k.  ok (nullable)
// marks the matrices whose state is already valid (accepted by launch_gen_inverse between two
// EM iterations); the others are filled from (eigvec, eigval) first.
inline int gen_state_ld(int D) { return D <= 12 ? 12 : (D + 3) / 4 * 4; }
inline size_t gen_state_doubles(int64_t N, int D) {
  return (size_t)N * gen_state_ld(D) * gen_state_ld(D) * 2;
}
struct GenInverseState {
  double* inv;           // c128 (N,LD,LD) workspace
  double* logdet;        // (N) workspace
  const int32_t* ok;     // (N) or null
};

// a2-a4: posteriors / quadratic form / log-pdf from an eigen-parameterised model
int launch_gen_estep(const void* y, int y_is_c128, int layout, int64_t B, int T, int D, int K,
                     const double* eigvec, const double* eigval, const double* weight, int64_t wb,
                     int64_t wk, int64_t wt, const uint8_t* activity, double eps, double* out_aff,
                     double* out_q, double* out_logpdf, hipStream_t s,
                     const GenInverseState& state, const double* saliency = nullptr,
                     double* out_mweight = nullptr, int32_t* out_zero = nullptr,
                     int raw_dt = 0);  // raw_dt: layout DT holds RAW values (transposed copy)

// (B, T, D) -> (B, D, T) copy of the raw observation for the E-steps of the EM loop
int launch_gen_transpose(const void* y, int y_is_c128, int64_t B, int T, int D, void* out,
                         hipStream_t s);

// EM loop, (B, T, D) observations: the M-step as three small steps around the E-step.
//   launch_gen_estep(..., saliency, out_mweight, out_zero) leaves the per-frame M-step weights
//   gamma sal / max(q, 10 tiny) / |y|^2 (cacg.py:310, :322) and flags bins with all-zero frames;
//   launch_gen_init_weights does the same for an affiliation initialisation (q = 1);
//   launch_gen_mstep_cov: class sums + mixture weights (a5), then C_k = D sum_t w y y^H / sum
//   (a6) on the DPP-operand kernel gen_cov2 (csum: (B,K) workspace).
int launch_gen_init_weights(const void* y, int y_is_c128, int layout, int64_t B, int T, int D, int K,
                            const double* gamma0, const double* saliency, double* out_mweight,
                            int32_t* out_zero, hipStream_t s);
int launch_gen_mstep_cov(const void* y, int y_is_c128, int64_t B, int T, int D, int K,
                         const double* mweight, const double* gamma, const double* saliency,
                         int weight_mode, double* csum, double* out_cov, double* out_weight,
                         hipStream_t s);

// a6 / a10: weighted covariances.  mode 0: M-step (D * sum_t gamma sal / q y y^H / sum gamma sal,
// observation unit-normalised when layout is TD); mode 1: PSD with the mask normalised by
// max(sum_t mask, 1e-10); mode 2: PSD plain sums (/T without a mask).  out_weight: mixture
// weights of the M-step (a5), out_sum: class sums.
int launch_gen_cov(const void* y, int y_is_c128, int layout, int64_t B, int T, int D, int K,
                   const double* gamma, int64_t gamma_bstride, const double* q,
                   const double* saliency, int mode, int weight_mode, double* out_cov,
                   double* out_weight, double* out_sum, size_t lds_limit, hipStream_t s,
                   int32_t* out_zero = nullptr);

// a7: Hermitian eigendecomposition (ascending, eigenvectors in columns).  covariance_norm < 0:
// plain numpy.linalg.eigh; otherwise the normalisation / floor of from_covariance.
// skip (nullable): matrices with skip[n] != 0 are left untouched (their inverse was accepted).
int launch_gen_heev(const double* a, int64_t N, int D, int covariance_norm, double eig_floor,
                    double* out_val, double* out_vec, int32_t* out_status, size_t lds_limit,
                    hipStream_t s, const int32_t* skip = nullptr);

// Fast path between EM iterations: in-place Gauss-Jordan inverse of the Hermitian positive
// definite covariance and its log-determinant.  The class log-pdf is invariant to the scale
// of B, so neither normalisation of cacg.py:82-132 matters here; only the eigenvalue floor
// does, and ok[n] is set only when tr(C) ||C^-1||_F (an upper bound of the condition number)
// proves that no eigenvalue would have been floored.  The invariance does not hold for an
// all-zero frame (its quadratic form is clamped, cacg.py:185-199, so the posterior depends on
// the normalised determinant): veto[n / K] != 0 (launch_gen_cov's out_zero) rejects every
// class of such a bin.
int launch_gen_inverse(const double* a, int64_t N, int D, double eig_floor, double* out_inv,
                       double* out_logdet, int32_t* out_ok, hipStream_t s,
                       const int32_t* veto = nullptr, int K = 1);

}  // namespace pbbss
