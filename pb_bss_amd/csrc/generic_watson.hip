// Complex-Watson mixture model at generic sizes (9 <= D <= 32 sensors or more than 4 classes):
// the pieces of one EM iteration that generic.hip does not already have.  Reference:
// pb_bss/distribution/complex_watson.py:73-88 (log_pdf), :157-168 (log_norm_1f1), :238-271
// (spline of the inverse hypergeometric ratio), :300-315 (_fit); cwmm.py:217-240 (_m_step).
// The fused kernel of cwmm.hpp serves D <= 8, K <= 4; here an iteration is
//   class log-pdfs (below) -> softmax with the weights (mixw.hip) -> gen_cov mode 3 (weights,
//   masked covariance of the unit-norm frames) -> gen_heev -> principal pair, kappa, ln c (below)
// enqueued back to back by pbbss_cwmm_fit.
#include "generic.hpp"
#include <cmath>
#include "pbbss_dev.hpp"

namespace pbbss {
namespace {

constexpr int kWT = 256;

// ln 1F1(1; D; kappa) = ln sum_m kappa^m / (D)_m: all terms positive, no cancellation; the terms
// grow until m ~ kappa - D and the tail is geometric afterwards (kappa <= max_concentration,
// 500 by default: ~700 terms at worst, ~40 for kappa < D)
__device__ double log_hyp1f1_1(int D, double kappa) {
  if (!(kappa > 0.0)) return 0.0;
  double s = 1.0, term = 1.0;
  for (int m = 1; m < 4000; ++m) {
    term *= kappa / (double)(D + m - 1);
    s += term;
    if (term < 1e-17 * s && (double)(D + m) > kappa) break;
  }
  return log(s);
}
__device__ double watson_log_norm_d(int D, double kappa) {
  return 0.6931471805599453 + (double)D * 1.1447298858494002 /* ln pi */ - lgamma((double)D) +
         log_hyp1f1_1(D, kappa);
}

// scipy.interpolate.interp1d(kind='quadratic', bounds_error=False, fill_value=(0, max)) ==
// BSpline(t, c, k=2) evaluated with de Boor inside [ev_min, ev_max] (one thread, binary search)
__device__ double watson_concentration_d(const GenWatsonSpline& a, double ev) {
  if (!(ev >= a.ev_min)) return 0.0;  // also NaN -> 0 like fill_value below the range
  if (ev > a.ev_max) return a.max_concentration;
  constexpr int k = 2;
  int lo = k, hi = a.n_coef - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (a.t[mid] <= ev) lo = mid; else hi = mid - 1;
  }
  const int i = lo;
  double d[k + 1];
  for (int j = 0; j <= k; ++j) d[j] = a.c[j + i - k];
  for (int r = 1; r <= k; ++r)
    for (int j = k; j >= r; --j) {
      const double tl = a.t[j + i - k], tr = a.t[j + 1 + i - r];
      const double alpha = (ev - tl) / (tr - tl);
      d[j] = (1.0 - alpha) * d[j - 1] + alpha * d[j];
    }
  return d[k];
}

// grid (frame tiles, bins), thread = frame: the frame's D channels are 8 D (16 D) contiguous
// bytes; the K modes of the bin sit in LDS and are read as broadcasts
template <typename YS>
__global__ void __launch_bounds__(kWT)
    gen_watson_logpdf_kernel(const YS* __restrict__ y, int T, int D, int K,
                             const double* __restrict__ mode, const double* __restrict__ conc,
                             const double* __restrict__ lognorm, double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* wm = reinterpret_cast<double*>(smem);  // [K][D][2]
  const int64_t b = blockIdx.y;
  const int tid = threadIdx.x;
  for (int e = tid; e < K * D * 2; e += kWT) wm[e] = mode[(size_t)b * K * D * 2 + e];
  __syncthreads();
  const int t = blockIdx.x * kWT + tid;
  if (t >= T) return;
  const YS* yp = y + ((size_t)b * T + t) * D * 2;
  double n2 = 0.0;
  for (int d = 0; d < D; ++d) {
    const double re = (double)yp[2 * d], im = (double)yp[2 * d + 1];
    n2 += re * re + im * im;
  }
  const double inv = (n2 > 0.0) ? 1.0 / n2 : 0.0;  // y / max(|y|, tiny): an all-zero frame stays 0
  for (int k = 0; k < K; ++k) {
    double sr = 0.0, si = 0.0;  // sum_d y_d conj(w_d)
    const double* w = wm + (size_t)k * D * 2;
    for (int d = 0; d < D; ++d) {
      const double re = (double)yp[2 * d], im = (double)yp[2 * d + 1];
      sr += re * w[2 * d] + im * w[2 * d + 1];
      si += im * w[2 * d] - re * w[2 * d + 1];
    }
    out[((size_t)b * K + k) * T + t] =
        conc[b * K + k] * ((sr * sr + si * si) * inv) - lognorm[b * K + k];
  }
}

__global__ void __launch_bounds__(kWT)
    gen_watson_lognorm_kernel(const double* __restrict__ conc, int64_t N, int D,
                              double* __restrict__ out) {
  const int64_t n = (int64_t)blockIdx.x * kWT + threadIdx.x;
  if (n < N) out[n] = watson_log_norm_d(D, conc[n]);
}

__global__ void __launch_bounds__(kWT)
    gen_watson_finish_kernel(const double* __restrict__ eigval, const double* __restrict__ eigvec,
                             int64_t N, int D, GenWatsonSpline sp, double* __restrict__ out_mode,
                             double* __restrict__ out_conc, double* __restrict__ out_lognorm) {
  const int64_t n = (int64_t)blockIdx.x * kWT + threadIdx.x;
  if (n >= N) return;
  // get_pca (utils.py:150-165): eigenvector of the largest eigenvalue (ascending order: the last)
  for (int d = 0; d < D; ++d) {
    out_mode[((size_t)n * D + d) * 2] = eigvec[(((size_t)n * D + d) * D + D - 1) * 2];
    out_mode[((size_t)n * D + d) * 2 + 1] = eigvec[(((size_t)n * D + d) * D + D - 1) * 2 + 1];
  }
  const double kappa = watson_concentration_d(sp, eigval[(size_t)n * D + D - 1]);
  out_conc[n] = kappa;
  out_lognorm[n] = watson_log_norm_d(D, kappa);
}

inline int ok_or_hip_w() { return hipGetLastError() == hipSuccess ? PBBSS_OK : PBBSS_ERR_HIP; }

}  // namespace

int launch_gen_watson_logpdf(const void* y, int y_is_c128, int64_t B, int T, int D, int K,
                             const double* mode, const double* conc, const double* lognorm,
                             double* out_logpdf, hipStream_t s) {
  if (!gen_supported(D, K)) return PBBSS_ERR_UNSUPPORTED;
  if (B > 65535) return PBBSS_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)((T + kWT - 1) / kWT), (unsigned)B);
  const size_t lds = (size_t)K * D * 2 * sizeof(double);
  if (y_is_c128)
    hipLaunchKernelGGL(gen_watson_logpdf_kernel<double>, grid, dim3(kWT), lds, s,
                       static_cast<const double*>(y), T, D, K, mode, conc, lognorm, out_logpdf);
  else
    hipLaunchKernelGGL(gen_watson_logpdf_kernel<float>, grid, dim3(kWT), lds, s,
                       static_cast<const float*>(y), T, D, K, mode, conc, lognorm, out_logpdf);
  return ok_or_hip_w();
}

int launch_gen_watson_lognorm(const double* conc, int64_t N, int D, double* out, hipStream_t s) {
  hipLaunchKernelGGL(gen_watson_lognorm_kernel, dim3((unsigned)((N + kWT - 1) / kWT)), dim3(kWT), 0,
                     s, conc, N, D, out);
  return ok_or_hip_w();
}

int launch_gen_watson_finish(const double* eigval, const double* eigvec, int64_t N, int D,
                             const GenWatsonSpline& sp, double* out_mode, double* out_conc,
                             double* out_lognorm, hipStream_t s) {
  hipLaunchKernelGGL(gen_watson_finish_kernel, dim3((unsigned)((N + kWT - 1) / kWT)), dim3(kWT), 0,
                     s, eigval, eigvec, N, D, sp, out_mode, out_conc, out_lognorm);
  return ok_or_hip_w();
}

}  // namespace pbbss
