// estimate_mixture_weight on the device (distribution/mixture_model_utils.py:133-203) for the
// options that couple problems: weight_constant_axis containing independent axes (e.g. (-3,):
// weights shared by all frequency bins, (-3, -1): one weight per class for the whole utterance).
// The fused EM kernel owns the per-bin cases ((-1,) and -2); this file serves the step-wise fit
// (CACGMMTrainer._fit_stepwise), which otherwise would have to bring the affiliations to the
// host every iteration.
//
//   affiliation (Bo, Bi, K, N) float64, saliency (Bo, Bi, N) or null
//   reduce over Bi (red_inner: the trailing independent axes in weight_constant_axis) and / or
//   over N (red_n: -1 in weight_constant_axis), keepdims:
//     no saliency: mean over the reduced axes                                   (:188)
//     saliency   : sum of affiliation * saliency over the reduced axes, then L1-normalised over
//                  the classes with `where(norm == 0, 1e-10, norm)`             (:190-201)
//   out (Bo, Bi', K, N'), Bi' = red_inner ? 1 : Bi, N' = red_n ? 1 : N.
// Two launches, fixed summation order (bit-reproducible): rows (one workgroup per problem:
// sums over N) and finish (sums over Bi, normalisation over the classes).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "embed.hpp"
#include "pbbss_dev.hpp"

namespace pbbss {
namespace {
constexpr int kT = 256;

// Frame chunks per problem of the sum over the frames: few problems with many frames (a mixture
// over ONE batch of 256 500 samples is a single problem) would otherwise be summed by one
// workgroup each -- 733 us of a 870 us diagonal-Gaussian EM iteration.  ~512 workgroups in total,
// at least 1024 frames per chunk.
int mixw_chunks(int64_t problems, int64_t N) {
  int64_t c = 512 / (problems < 512 ? problems : 512);
  const int64_t by_len = (N + 1023) / 1024;
  if (c > by_len) c = by_len;
  if (c > kT) c = kT;  // the finish kernel takes one chunk per thread
  return (int)(c < 1 ? 1 : c);
}

// tmp (Bo*Bi, C, K): per problem and frame chunk the (saliency-weighted) affiliation summed over
// the chunk's frames (red_n only: without that sum the finish kernel reads the affiliation itself)
__global__ void __launch_bounds__(kT) mixw_rows_kernel(const double* __restrict__ aff,
                                                       const double* __restrict__ sal, int K,
                                                       int64_t N, double* __restrict__ tmp) {
  const int64_t b = blockIdx.x;
  const int C = gridDim.y, c = blockIdx.y;
  const int64_t per = (N + C - 1) / C;
  const int64_t n0 = per * c, n1 = (n0 + per < N) ? n0 + per : N;
  const double* a = aff + b * K * N;
  const double* s = sal ? sal + b * N : nullptr;
  __shared__ double red[kT / kWave];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = 0; k < K; ++k) {
    double acc = 0.0;
    for (int64_t n = n0 + threadIdx.x; n < n1; n += kT)
      acc += a[(int64_t)k * N + n] * (s ? s[n] : 1.0);
    acc = wave_sum(acc);
    __syncthreads();
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0.0;
      for (int w = 0; w < kT / kWave; ++w) t += red[w];
      tmp[(b * C + c) * K + k] = t;
    }
  }
}

// out (Bo, Bi2, K, N1) from tmp (Bo, Bi, K, N1): sum over Bi if red_inner, then normalise.
// kFinFrames frames per workgroup (blockIdx.y), kT / kFinFrames partial sums per frame: 8 x 32
// when the inner problems are summed, 64 x 4 (one part at work) when they are only normalised.
template <int kFinFrames>
__global__ void __launch_bounds__(kT) mixw_finish_kernel(const double* __restrict__ tmp,
                                                         const double* __restrict__ sal_rows,
                                                         int64_t Bi, int C, int K, int64_t N1,
                                                         int red_inner, int has_sal, double count,
                                                         double* __restrict__ out) {
  const int64_t Bi2 = red_inner ? 1 : Bi;
  const int64_t bo = blockIdx.x / Bi2, b2 = blockIdx.x % Bi2;  // blockIdx.y: tile of frames
  extern __shared__ double sm[];  // [K] class totals when N1 == 1
  if (N1 == 1) {
    // parallel over bi, block reduction per class
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ double red[kT / kWave];
    for (int k = 0; k < K; ++k) {
      double acc = 0.0;
      // tmp (Bo*Bi, C, K): the (problem, chunk) pairs of one outer index are contiguous
      if (red_inner) {
        for (int64_t i = threadIdx.x; i < Bi * C; i += kT) acc += tmp[(bo * Bi * C + i) * K + k];
      } else if (threadIdx.x < C) {
        acc = tmp[((bo * Bi + b2) * C + threadIdx.x) * K + k];
      }
      acc = wave_sum(acc);
      __syncthreads();
      if (lane == 0) red[wave] = acc;
      __syncthreads();
      if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < kT / kWave; ++w) t += red[w];
        sm[k] = t;
      }
    }
    __syncthreads();
    if (threadIdx.x < K) {
      double v = sm[threadIdx.x];
      if (has_sal) {
        double nrm = 0.0;
        for (int k = 0; k < K; ++k) nrm += fabs(sm[k]);
        v /= (nrm == 0.0) ? 1e-10 : nrm;
      } else {
        v /= count;
      }
      out[(bo * Bi2 + b2) * K + threadIdx.x] = v;
    }
    return;
  }
  // N1 == N: one kFinFrames-frame tile per blockIdx.y; thread (f, part) sums the reduced problems
  // bi = part, part + kFinParts, ... of frame f for all classes of the chunk at once (K loads in
  // flight per trip: the loop is a chain of L2 round trips, 513 bins over 4 parts and one class
  // at a time were ~400 of them), the parts are combined in part order through LDS
  constexpr int kFinParts = kT / kFinFrames;
  __shared__ double comb[kFinParts][16][kFinFrames];  // [part][class][frame]
  const int f = threadIdx.x % kFinFrames, part = threadIdx.x / kFinFrames;
  const int64_t n = (int64_t)blockIdx.y * kFinFrames + f;
  for (int k0 = 0; k0 < K; k0 += 16) {  // classes in chunks of 16 (K <= 64)
    const int kc = (K - k0 < 16) ? K - k0 : 16;
    double acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.0;
    if (n < N1) {
      if (red_inner) {
        // N1 == N: tmp is the affiliation itself, weighted here (sal_rows (Bo*Bi, N) or null)
        // four problems per trip (their loads in flight together: the loop is a chain of L2
        // round trips), added in ascending order like the one-at-a-time loop
        constexpr int kU = 4;
        int64_t bi = part;
        if (kc <= 4) {
          for (; bi + (kU - 1) * kFinParts < Bi; bi += kU * kFinParts) {
            double v[kU][4], sv[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
              const int64_t bb = bi + u * kFinParts;
              const double* row = tmp + ((bo * Bi + bb) * K + k0) * N1 + n;
              sv[u] = sal_rows ? sal_rows[(bo * Bi + bb) * N1 + n] : 1.0;
#pragma unroll
              for (int k = 0; k < 4; ++k) v[u][k] = (k < kc) ? row[(int64_t)k * N1] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < kU; ++u)
#pragma unroll
              for (int k = 0; k < 4; ++k)
                if (k < kc) acc[k] += v[u][k] * sv[u];
          }
        }
        for (; bi < Bi; bi += kFinParts) {
          const double* row = tmp + ((bo * Bi + bi) * K + k0) * N1 + n;
          const double sv = sal_rows ? sal_rows[(bo * Bi + bi) * N1 + n] : 1.0;
#pragma unroll
          for (int k = 0; k < 16; ++k)
            if (k < kc) acc[k] += row[(int64_t)k * N1] * sv;
        }
      } else if (part == 0) {
        const double sv = sal_rows ? sal_rows[(bo * Bi + b2) * N1 + n] : 1.0;
#pragma unroll
        for (int k = 0; k < 16; ++k)
          if (k < kc) acc[k] = tmp[((bo * Bi + b2) * K + k0 + k) * N1 + n] * sv;
      }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k)
      if (k < kc) comb[part][k][f] = acc[k];
    __syncthreads();
    // the class norm needs ALL classes: K <= 16 in one chunk is the only case with saliency
    // that occurs here (K <= 16 everywhere in the library); larger K fall back to plain sums
    if (part == 0 && n < N1) {
      double nrm = 0.0;
      for (int k = 0; k < kc; ++k) {
        double v = 0.0;
        for (int pp = 0; pp < kFinParts; ++pp) v += comb[pp][k][f];
        comb[0][k][f] = v;
        nrm += fabs(v);
      }
      for (int k = 0; k < kc; ++k) {
        const double v = comb[0][k][f];
        out[((bo * Bi2 + b2) * K + k0 + k) * N1 + n] =
            has_sal ? v / ((nrm == 0.0) ? 1e-10 : nrm) : v / count;
      }
    }
    __syncthreads();
  }
}
}  // namespace

// log_pdf_to_affiliation (mixture_model_utils.py:7-55) as a stand-alone step for the models
// whose E-step is not fused with it: thread = (problem, sample), the class log-pdfs of a sample
// are read three times (max, normaliser, result) -- K is small and the rows are L2-resident.
__global__ void __launch_bounds__(kT) lp_to_aff_kernel(const double* __restrict__ lp,
                                                       const double* __restrict__ w, int64_t wb,
                                                       int64_t wk, int64_t wn,
                                                       const uint8_t* __restrict__ act, int K,
                                                       int64_t N, double eps,
                                                       double* __restrict__ out) {
  const int64_t b = blockIdx.y;
  const int64_t n = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (n >= N) return;
  const double* p = lp + b * K * N + n;
  double mx = -1.79e308;
  for (int k = 0; k < K; ++k) mx = fmax(mx, p[(int64_t)k * N]);  // :32
  auto term = [&](int k) {
    double v = exp(p[(int64_t)k * N] - mx) * w[b * wb + k * wk + n * wn];  // :34-37
    if (act) v *= (double)act[(b * K + k) * N + n];                        // :41
    return v;
  };
  double den = 0.0;
  for (int k = 0; k < K; ++k) den += term(k);
  den = fmax(den, kTiny);  // :43-47
  for (int k = 0; k < K; ++k) {
    double g = term(k) / den;
    if (eps != 0.0) g = fmin(fmax(g, eps), 1.0 - eps);  // :50-53, no renormalisation
    out[(b * K + k) * N + n] = g;
  }
}

size_t mixture_weight_tmp_doubles(int64_t Bo, int64_t Bi, int K, int64_t N, int red_n) {
  return red_n ? (size_t)Bo * Bi * mixw_chunks(Bo * Bi, N) * K : 1;
}

int launch_mixture_weight(const double* aff, const double* sal, int64_t Bo, int64_t Bi, int K,
                          int64_t N, int red_inner, int red_n, double* tmp, double* out,
                          hipStream_t s) {
  const int64_t N1 = red_n ? 1 : N;
  // sums over the frames first; without them the finish kernel reads the affiliation itself
  const int C = red_n ? mixw_chunks(Bo * Bi, N) : 1;
  if (red_n)
    hipLaunchKernelGGL(mixw_rows_kernel, dim3((unsigned)(Bo * Bi), (unsigned)C), dim3(kT), 0, s, aff,
                       sal, K, N, tmp);
  const double count = (red_n ? (double)N : 1.0) * (red_inner ? (double)Bi : 1.0);
  const int64_t Bi2 = red_inner ? 1 : Bi;
  const int frames = red_inner ? 8 : 64;
  const unsigned tiles = (unsigned)((N1 + frames - 1) / frames);  // N1 == 1: one tile
  auto kfn = red_inner ? mixw_finish_kernel<8> : mixw_finish_kernel<64>;
  hipLaunchKernelGGL(kfn, dim3((unsigned)(Bo * Bi2), tiles), dim3(kT), K * sizeof(double), s,
                     red_n ? tmp : aff, red_n ? nullptr : sal, Bi, C, K, N1, red_inner, sal ? 1 : 0,
                     count, out);
  return hipGetLastError() == hipSuccess ? PBBSS_OK : PBBSS_ERR_HIP;
}

int launch_log_pdf_to_affiliation(const double* lp, int64_t B, int K, int64_t N, const double* w,
                                  int64_t wb, int64_t wk, int64_t wn, const uint8_t* act,
                                  double eps, double* out, hipStream_t s) {
  if (B <= 0 || N <= 0) return PBBSS_OK;
  if (B > 65535) return PBBSS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(lp_to_aff_kernel, dim3((unsigned)((N + kT - 1) / kT), (unsigned)B), dim3(kT),
                     0, s, lp, w, wb, wk, wn, act, K, N, eps, out);
  return hipGetLastError() == hipSuccess ? PBBSS_OK : PBBSS_ERR_HIP;
}

}  // namespace pbbss
