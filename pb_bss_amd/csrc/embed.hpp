// Real-embedding mixture components: the von
// Mises-Fisher and spherical-Gaussian halves of pb_bss.distribution.{vmfmm,
// gcacgmm, vmfcacgmm}.  Host-side launchers; kernels live in embed.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "pbbss.h"

namespace pbbss {

constexpr int kEmbedMaxK = 8;     // classes (the fused spatial kernels take 6; 7 and 8 pair with the generic-size path)
constexpr int kEmbedMaxE = 256;   // embedding dimension (one workgroup row of the fit kernel)

// Sharded fits (the points of ONE mixture spread over several GPUs): called between the kernel
// that writes per-chunk partial sums and the kernel that finalises them, to sum the partials over
// the ranks in place (capi.hip binds it to an ncclAllReduce on the launch stream).
struct PartialReduce {
  int (*fn)(void* ctx, double* buf, size_t count, hipStream_t s);
  void* ctx;
};

// bytes of the (B, E, N) transposed copy / of the fit partial sums
size_t embed_partial_doubles(int64_t B, int64_t N, int E, int K, int* chunks_out);

// y (B,N,E) row-major -> yd (B,E,N); NORMALIZE: rows scaled to unit norm
// (von_mises_fisher.py:105-107) and ALSO written row-major to yr (float64).
// normalize = 0 with `rowscale` (B,N): raw copy plus 1 / max(|y_n|, tiny) per row, for callers
// that apply the unit-norm scaling on the fly (the vMF mixture loop).
int launch_embed_prepare(const void* y, int y_is_f64, int64_t B, int64_t N, int E, int normalize,
                         void* yd, double* yr, hipStream_t s, double* rowscale = nullptr);

// offset_k of the class log-pdfs: vMF  -log_norm(kappa)        (von_mises_fisher.py:33-44)
//                                 Gauss -E/2 ln 2pi + E ln(1/sqrt(cov)) (gaussian.py:108-137)
// and prec_k: vMF kappa; Gauss 1/cov.
int launch_embed_offsets(int kind, int64_t BK, int E, const double* scale, double* offset,
                         double* prec, hipStream_t s);

// class log-pdfs (times out_scale) and/or posteriors of every sample.
//   yd (B,E,N) of type y_is_f64; mean (B,K,E); prec/offset (B,K) from launch_embed_offsets;
//   weight (B,K) or null (needed for out_aff); out index of (b,k,n):
//   b*K*N + (n / Tin)*K*Tin + k*Tin + n % Tin   (Tin = N: plain (B,K,N); Tin = T: 'k,ft->fkt')
int launch_embed_estep(int kind, const void* yd, int y_is_f64, int64_t B, int64_t N, int E, int K,
                       const double* mean, const double* prec, const double* offset,
                       const double* weight, double out_scale, int64_t Tin, double* out_lp,
                       double* out_aff, hipStream_t s);

// weighted fit:  w_k(n) = aff[index(b,k,n)] * (sal ? sal[b*N+n] : 1)
//   vMF   (von_mises_fisher.py:119-144): mean direction, concentration clipped to [cmin,cmax]
//   Gauss (gaussian.py:152-193, 'spherical'): mean, then variance about that mean (2nd pass)
//   weight_mode >= 0 also writes the mixture weights (B,K) (mixture_model_utils.py:133-203):
//   0 L1-normalised masked sums, 1 uniform 1/K.
// part: scratch of embed_partial_doubles() doubles.  out_offset / out_prec (B,K), optional:
// what launch_embed_offsets would compute from out_scale, produced in the same launch.
// single_pass (spherical Gaussian only): 0 = two sweeps as the reference (mean, then variance
// about it); 1 = one sweep with the variance accumulated about out_mean's CURRENT content (the
// previous iteration's mean) and corrected in the finalize; 2 = same with the first row of y
// as the shift (no previous mean yet).
int launch_embed_fit(int kind, const void* yr, int y_is_f64, int64_t B, int64_t N, int E, int K,
                     const double* aff, int64_t Tin, const double* sal, double cmin, double cmax,
                     int weight_mode, double* part, double* out_mean, double* out_scale,
                     double* out_weight, double* out_offset, double* out_prec, int single_pass,
                     hipStream_t s, const double* rowscale = nullptr,  // rowscale: vMF only, rows
                     const PartialReduce* reduce = nullptr);           // are used as y_n * rowscale[n]

// ONE EM iteration of the vMF mixture in one pass over the row-major embedding (vmf_em_kernel:
// E-step and M-step partial sums on the same LDS tile) + the ordered finalize.  gamma (B,K,N):
// affiliations of the first iteration (then mean / prec / offset / weight are not read), or
// null: the E-step uses the current model (mean (B,K,E), prec = concentration, offset =
// -log_norm, weight (B,K)).  out_aff (B,K,N) or null: the affiliations of this sweep.
// accumulate = 0: E-step only (predict).  part: vmf_fused_partial_doubles() doubles (0: the shape
// is not served -- E > 256 or the tile does not fit -- and the caller uses the two-kernel path).
size_t vmf_fused_partial_doubles(int64_t B, int64_t N, int E, int K, int y_is_f64);
int launch_vmf_em(const void* y, int y_is_f64, int64_t B, int64_t N, int E, int K,
                  const double* gamma, const double* sal, double cmin, double cmax,
                  int weight_mode, double* part, double* mean, double* conc, double* weight,
                  double* offset, double* prec, double* out_aff, int accumulate, hipStream_t s);

// Persistent vMF mixture EM for MANY SMALL mixtures (round 4): one workgroup per mixture keeps its
// rows and the model in LDS for all iterations (vmf_bin_em_kernel).  gamma (B,K,N): affiliation
// initialisation (iterations > 0), else the model (in_mean (B,K,E), in_conc, in_weight (B,K)) for a
// pure predict (iterations == 0).  out_aff (B,K,N) or null.  vmf_bin_lds_bytes: dynamic LDS the
// shape needs (0: not served).
size_t vmf_bin_lds_bytes(int64_t N, int E, int K, int y_is_f64);
int launch_vmf_bin_em(const void* y, int y_is_f64, int64_t B, int64_t N, int E, int K,
                      int iterations, const double* gamma, const double* sal,
                      const double* in_mean, const double* in_conc, const double* in_weight,
                      double cmin, double cmax, int weight_mode, double* mean, double* conc,
                      double* weight, double* out_aff, size_t lds_limit, hipStream_t s);

// Round-6 kernel for the same job at E <= 16 features, 2 <= K <= 4 classes (vmf_bin.hip: feature
// count as a template parameter, E and M in one sweep with lane = row, class means as DPP
// operands, eight wavefronts per mixture).  PBBSS_ERR_UNSUPPORTED: shape not served, take
// launch_vmf_bin_em.
int launch_vmf_bin_em2(const void* y, int y_is_f64, int64_t B, int64_t N, int E, int K,
                       int iterations, const double* gamma, const double* sal,
                       const double* in_mean, const double* in_conc, const double* in_weight,
                       double cmin, double cmax, int weight_mode, double* mean, double* conc,
                       double* weight, double* out_aff, size_t lds_limit, int num_cu,
                       hipStream_t s);

// Rotated joint loop (round 4): ONE pass over the row-major embedding (F*T, E) per EM iteration of
// GCACGMM (spherical) / VMFCACGMM.  launch_joint_sweep: posteriors of every point from the
// spatial quadratic forms Q (F,K,T) + ln det B (F,K) (written by the spatial kernel,
// cacgmm_em.hpp: run_joint_ms) and the spectral model (mean (K,E), prec / offset (K)) -> G (F,K,T)
// (clipped to [eps, 1 - eps]); chunk partials of the spectral M-step sums with gamma * saliency
// from the same tile -> part (joint_sweep_partial_doubles() doubles).  weight is read through
// (wb, wk, wt) as w[f * wb + k * wk + t * wt] (null: 1).  launch_joint_sweep_finalize: the
// spectral model of the next iteration from the partials (all-reduced first when `reduce`).
bool joint_sweep_supported(int kind, int64_t N, int E, int K, int y_is_f64);
size_t joint_sweep_partial_doubles(int kind, int64_t N, int E, int K, int y_is_f64);
void joint_sweep_chunks(int kind, int64_t N, int E, int K, int y_is_f64, int* chunks);
int launch_joint_sweep(int kind, const void* y, int y_is_f64, int64_t F, int T, int E, int K, int D,
                       const double* Q, const double* lndet, const double* weight, int64_t wb,
                       int64_t wk, int64_t wt, const double* mean, const double* prec,
                       const double* offset, double spatial_weight, double spectral_weight,
                       const double* sal, double eps, double* G, double* part, hipStream_t s);
int launch_joint_sweep_finalize(int kind, const void* y, int y_is_f64, int64_t N, int E, int K,
                                double cmin, double cmax, double* part, double* mean,
                                double* scale, double* offset, double* prec, hipStream_t s,
                                const PartialReduce* reduce = nullptr);

// masked affiliation sums of the joint models (gcacgmm.py:286-295): aff (F,K,T), sal (F,T)
//   mode 0 'fk' (-1,): w[f,k] = sum_t / sum_k sum_t          -> (F,K)
//   mode 1 uniform    : 1/K                                   -> (1)
//   mode 2 'k' (-3,-1): sum_ft / sum_k sum_ft                 -> (K)
//   mode 3 'kt' (-3,) : sum_f / sum_k sum_f                   -> (K,T)
//   mode 4 ''         : 1                                     -> (1)
// tmp: joint_weight_tmp_doubles(mode, F, K, T) doubles of scratch.
// reduce != null (bins sharded over ranks): the sums of modes 2 / 3 are all-reduced before they
// are normalised over the classes.
size_t joint_weight_tmp_doubles(int mode, int64_t F, int K, int T);
int launch_joint_weight(int mode, const double* aff, const double* sal, int64_t F, int K, int T,
                        double* tmp, double* out_weight, hipStream_t s,
                        const PartialReduce* reduce = nullptr);

// DiagonalGaussian.log_pdf as the reference writes it (gaussian.py:76-97, see embed.hip):
// yd (E, N) transposed copy, mean / cov (K, E); out index as launch_embed_estep (b = 0);
// consts: diag_consts_doubles() of scratch.
size_t diag_consts_doubles(int K, int E);
int launch_diag_estep(const void* yd, int y_is_f64, int64_t N, int E, int K, const double* mean,
                      const double* cov, double out_scale, int64_t Tin, double* consts,
                      double* out_lp, hipStream_t s);
// first row of a real (N, E) array as float64 (the common shift of a sharded full-covariance fit)
int launch_first_row_f64(const void* y, int y_is_f64, int E, double* out, hipStream_t s);
// (F, K, T) <-> (K, F*T) for the full-covariance kernels of gauss_full.hip
int launch_fkt_to_kn(const double* aff, const double* sal, int64_t F, int K, int T, double* out,
                     hipStream_t s);
int launch_kn_to_fkt(const double* lp, double scale, int64_t F, int K, int T, double* out,
                     hipStream_t s);

// dst[0] |= src[0] (spectral-half status of the joint fit)
int launch_or_status(const int32_t* src, int32_t* dst, hipStream_t s);

// mixw.hip: estimate_mixture_weight (mixture_model_utils.py:133-203) with reductions over
// independent axes; affiliation (Bo, Bi, K, N), saliency (Bo, Bi, N) or null -> out
// (Bo, red_inner ? 1 : Bi, K, red_n ? 1 : N); tmp: mixture_weight_tmp_doubles() of scratch.
size_t mixture_weight_tmp_doubles(int64_t Bo, int64_t Bi, int K, int64_t N, int red_n);
int launch_mixture_weight(const double* aff, const double* sal, int64_t Bo, int64_t Bi, int K,
                          int64_t N, int red_inner, int red_n, double* tmp, double* out,
                          hipStream_t s);

// mixw.hip: log_pdf_to_affiliation (mixture_model_utils.py:7-55): lp (B,K,N); the weight of
// (b,k,n) at w[b*wb + k*wk + n*wn] (a zero stride broadcasts); act (B,K,N) uint8 or null.
int launch_log_pdf_to_affiliation(const double* lp, int64_t B, int K, int64_t N, const double* w,
                                  int64_t wb, int64_t wk, int64_t wn, const uint8_t* act,
                                  double eps, double* out, hipStream_t s);

}  // namespace pbbss
