// Generic-size (9 <= D <= 32 sensors) cACGMM kernels: host-callable launchers (generic.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pbbss.h"

namespace pbbss {

constexpr int kGenMaxD = 32;
// the cACGMM trainer's own kernels (E-step, covariance, eigendecomposition, inverse) also serve
// D = 33, 34: the reference's sanity assert admits D < 35 (cacgmm.py:250); 36-wide padded tiles
constexpr int kGenMaxDEm = 34;

bool gen_supported(int D, int K);
bool gen_em_supported(int D, int K);  // the cACGMM fit / predict path: D <= kGenMaxDEm

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
                     int raw_dt = 0,  // raw_dt: layout DT holds RAW values (transposed copy)
                     // joint spatial + spectral models: posterior exponent = spatial_scale *
                     // log-pdf + extra[b,k,t] (gcacgmm.py:66-117)
                     const double* extra = nullptr, double spatial_scale = 1.0);

// Joint models with inline permutation alignment (mixture_model_utils.py:58-130): from the
// spatial log-pdf / quadratic forms of launch_gen_estep (out_logpdf, out_q) and the spectral
// log-pdf `extra`, the best class permutation per bin, the posterior and the cACG M-step weights.
// yt: the (B, D, T) copy of the raw observation.  K <= 6.
int launch_gen_joint_pa(const void* yt, int y_is_c128, int64_t B, int T, int D, int K,
                        const double* lp_spatial, const double* q, const double* extra,
                        double spatial_scale, const double* weight, int64_t wb, int64_t wk,
                        int64_t wt, const double* saliency, double eps, double* out_aff,
                        double* out_mweight, int32_t* out_zero, hipStream_t s,
                        const uint8_t* activity = nullptr, int32_t* out_perm = nullptr);

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
// max(sum_t mask, 1e-10); mode 2: PSD plain sums (/T without a mask); mode 3: Watson M-step
// (complex_watson.py:300-315 with the mask gamma sal of cwmm.py:217-240: unit-norm frames when
// layout is TD, divided by the class sum).  out_weight: mixture weights of the M-step (a5),
// out_sum: class sums.
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

// ---- complex-Watson mixture at generic sizes (generic_watson.hip; cwmm.py, complex_watson.py)
struct GenWatsonSpline {
  const double* t;  // knots (n_coef + 3), device
  const double* c;  // coefficients (n_coef), device
  int n_coef;
  double ev_min, ev_max, max_concentration;
};
// class log-pdfs kappa |w^H y|^2 / |y|^2 - ln c(kappa) (complex_watson.py:73-88) of the raw
// (B, T, D) observation; mode c128 (B,K,D), conc / lognorm (B,K); out (B,K,T)
int launch_gen_watson_logpdf(const void* y, int y_is_c128, int64_t B, int T, int D, int K,
                             const double* mode, const double* conc, const double* lognorm,
                             double* out_logpdf, hipStream_t s);
// ln c(kappa) = ln(2 pi^D / (D-1)! 1F1(1; D; kappa)) (complex_watson.py:157-168)
int launch_gen_watson_lognorm(const double* conc, int64_t N, int D, double* out, hipStream_t s);
// principal eigenpair (ascending eigh output) -> mode, concentration through the inverse
// hypergeometric-ratio spline (complex_watson.py:238-271, :314), ln c of it
int launch_gen_watson_finish(const double* eigval, const double* eigvec, int64_t N, int D,
                             const GenWatsonSpline& sp, double* out_mode, double* out_conc,
                             double* out_lognorm, hipStream_t s);

}  // namespace pbbss
