// Real-embedding mixture kernels for gfx950.
//
// Reference: distribution/von_mises_fisher.py:33-44 (log_norm), :62-78 (log_pdf),
// :119-144 (fit); distribution/gaussian.py:108-137 (SphericalGaussian.log_pdf),
// :152-193 (GaussianTrainer._fit); distribution/mixture_model_utils.py:7-55, 133-203;
// distribution/gcacgmm.py:286-295 (class weights of the joint models).
//
// These paths stream N x E real embeddings (N = F*T ~ 2.6e5, E ~ 40) twice per EM
// iteration and do ~K flops per element: HBM/Infinity-Cache bound.  Layouts:
//   E-step  thread = sample, reads the (E, N) transposed copy: every load of a wave is
//           one contiguous 256/512-byte row segment; class means are LDS broadcasts.
//   M-step  thread = (sample slot s, dimension d) over the row-major (N, E) array: the
//           256 threads of a workgroup read one contiguous block of 256/E samples per
//           trip; per-thread accumulators for all classes, one LDS reduction over the
//           slots per workgroup, ordered (deterministic) reduction over workgroups in
//           a finalize kernel.
#include "embed.hpp"
#include <cmath>
#include "pbbss_dev.hpp"
#include "embed_dev.hpp"

namespace pbbss {
namespace {

constexpr int kThreads = 256;
constexpr double kLn2Pi = 1.8378770664093454;  // ln(2 pi)

__device__ __forceinline__ size_t aff_index(int64_t b, int k, int64_t n, int K, int64_t N,
                                            int64_t Tin) {
  const int64_t f = n / Tin;
  return (size_t)b * K * N + (size_t)f * K * Tin + (size_t)k * Tin + (size_t)(n - f * Tin);
}

// ---------------------------------------------------------------- prepare
// One workgroup: R rows of one mixture staged in LDS (row stride E+1), optional
// unit-norm scaling, written out transposed (and row-major when normalising).
// NORMALIZE = false with `rowscale`: the copy stays raw and 1 / max(|y_n|, tiny) goes to
// rowscale[b][n] -- the vMF mixture applies it on the fly instead of reading float64 copies.
template <typename TS, bool NORMALIZE>
__global__ void __launch_bounds__(kThreads) embed_prepare_kernel(const TS* y, int64_t N, int E,
                                                                 int R, void* yd_, double* yr,
                                                                 double* rowscale) {
  using OUT = typename std::conditional<NORMALIZE, double, TS>::type;
  extern __shared__ double sm[];
  double* tile = sm;                // [R][E+1]
  double* scale = sm + R * (E + 1);  // [R]
  const int64_t b = blockIdx.y;
  const int64_t n0 = (int64_t)blockIdx.x * R;
  const int rows = (int)((N - n0 < R) ? (N - n0) : R);
  const int tid = threadIdx.x;
  const TS* src = y + ((size_t)b * N + n0) * E;
  for (int i = tid; i < rows * E; i += kThreads) {
    int r = i / E, d = i - r * E;
    tile[r * (E + 1) + d] = (double)src[i];
  }
  __syncthreads();
  if (NORMALIZE) {
    if (tid < rows) {
      double n2 = 0.0;
      for (int d = 0; d < E; ++d) n2 = fma(tile[tid * (E + 1) + d], tile[tid * (E + 1) + d], n2);
      scale[tid] = 1.0 / fmax(sqrt(n2), kTiny);  // vmfmm.py:76-78
    }
    __syncthreads();
    double* dst = yr + ((size_t)b * N + n0) * E;
    for (int i = tid; i < rows * E; i += kThreads) {
      int r = i / E, d = i - r * E;
      dst[i] = tile[r * (E + 1) + d] * scale[r];
    }
  }
  if (!NORMALIZE && rowscale && tid < rows) {
    double n2 = 0.0;
    for (int d = 0; d < E; ++d) n2 = fma(tile[tid * (E + 1) + d], tile[tid * (E + 1) + d], n2);
    rowscale[(size_t)b * N + n0 + tid] = 1.0 / fmax(sqrt(n2), kTiny);  // vmfmm.py:76-78
  }
  OUT* yd = static_cast<OUT*>(yd_) + (size_t)b * E * N + n0;
  for (int i = tid; i < rows * E; i += kThreads) {
    int d = i / rows, r = i - d * rows;
    double v = tile[r * (E + 1) + d];
    if (NORMALIZE) v *= scale[r];
    yd[(size_t)d * N + r] = (OUT)v;
  }
}

// ---------------------------------------------------------------- offsets
// ln( I_nu(x) / x^nu ) by the ascending series (all terms positive), summed in the
// log domain by one wavefront:  sum_m (x^2/4)^m / (m! Gamma(m+nu+1)) * 2^-nu.
// (wave_log_bessel_over_power lives in embed_dev.hpp: the spatial kernel of the rotated joint loop
// runs the spectral finalize inside its own launch)

// one wavefront per (mixture, class)
__global__ void __launch_bounds__(kThreads) embed_offsets_kernel(int kind, int64_t BK, int E,
                                                                 const double* scale,
                                                                 double* offset, double* prec) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t i = (int64_t)blockIdx.x * (kThreads / kWave) + (threadIdx.x >> 6);
  if (i >= BK) return;
  const double sc = scale[i];
  if (kind == PBBSS_EMBED_VMF) {
    // log_norm = E/2 ln 2pi + ln ive(nu, k) + (|k| - nu ln k) = E/2 ln 2pi + ln(I_nu(k) / k^nu)
    const double nu = 0.5 * E - 1.0;
    const double ln_c = 0.5 * E * kLn2Pi + wave_log_bessel_over_power(nu, sc, lane);
    if (lane == 0) {
      offset[i] = -ln_c;
      prec[i] = sc;
    }
  } else if (lane == 0) {
    const double pc = 1.0 / sqrt(sc);  // sklearn _compute_precision_cholesky, 'diag' branch
    offset[i] = -0.5 * E * kLn2Pi + (double)E * log(pc);
    prec[i] = pc;
  }
}

// ---------------------------------------------------------------- E-step
template <int KIND, int K, typename TS>
__global__ void __launch_bounds__(kThreads)
    embed_estep_kernel(const TS* yd, int64_t N, int E, const double* mean, const double* prec,
                       const double* offset, const double* weight, double out_scale, int64_t Tin,
                       double* out_lp, double* out_aff) {
  extern __shared__ double sm[];
  double* mu = sm;            // [K][E]
  double* pr = sm + K * E;    // [K]
  double* of = pr + K;        // [K]
  double* wg = of + K;        // [K]
  const int64_t b = blockIdx.y;
  const int tid = threadIdx.x;
  for (int i = tid; i < K * E; i += kThreads) mu[i] = mean[(size_t)b * K * E + i];
  if (tid < K) {
    pr[tid] = prec[b * K + tid];
    of[tid] = offset[b * K + tid];
    wg[tid] = weight ? weight[b * K + tid] : 1.0;
  }
  __syncthreads();
  const int64_t n = (int64_t)blockIdx.x * kThreads + tid;
  if (n >= N) return;
  const TS* col = yd + (size_t)b * E * N + n;
  double acc[K], n2 = 0.0, pk[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    acc[k] = 0.0;
    pk[k] = pr[k];
  }
  constexpr int U = 8;  // loads in flight per lane (16 measured: no gain)
  for (int d0 = 0; d0 < E; d0 += U) {
    // unconditional loads at clamped rows, masked afterwards: a guarded load whose value is
    // converted inside the guard compiles to branch + load + s_waitcnt vmcnt(0) -- the eight
    // "loads in flight" were eight serial memory round trips
    TS raw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) raw[u] = col[(size_t)((d0 + u < E) ? d0 + u : E - 1) * N];
    double v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = (d0 + u < E) ? (double)raw[u] : 0.0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (d0 + u < E) {
        if (KIND == PBBSS_EMBED_VMF) {
          n2 = fma(v[u], v[u], n2);
#pragma unroll
          for (int k = 0; k < K; ++k) acc[k] = fma(v[u], mu[k * E + d0 + u], acc[k]);
        } else {
#pragma unroll
          for (int k = 0; k < K; ++k) {
            double w = pk[k] * (v[u] - mu[k * E + d0 + u]);  // gaussian.py:127-131
            acc[k] = fma(w, w, acc[k]);
          }
        }
      }
    }
  }
  double lp[K], mx = -1.79e308;
  const double inv = (KIND == PBBSS_EMBED_VMF) ? 1.0 / fmax(sqrt(n2), kTiny) : 0.0;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    lp[k] = (KIND == PBBSS_EMBED_VMF) ? fma(pk[k], acc[k] * inv, of[k])   // von_mises_fisher.py:71-77
                                      : of[k] - 0.5 * acc[k];            // gaussian.py:132-136
    mx = fmax(mx, lp[k]);
  }
  if (out_lp) {
#pragma unroll
    for (int k = 0; k < K; ++k) out_lp[aff_index(b, k, n, K, N, Tin)] = out_scale * lp[k];
  }
  if (out_aff) {  // mixture_model_utils.py:30-47, affiliation_eps = 0
    double g[K], den = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      g[k] = exp(lp[k] - mx) * wg[k];
      den += g[k];
    }
    den = fmax(den, kTiny);
#pragma unroll
    for (int k = 0; k < K; ++k) out_aff[aff_index(b, k, n, K, N, Tin)] = g[k] / den;
  }
}

// ---------------------------------------------------------------- M-step partial sums
// PASS 0: S1[k][d] = sum_n w_k(n) y[n][d], S0[k] = sum_n w_k(n)
// PASS 1: S2[k][d] = sum_n w_k(n) (y[n][d] - mean[k][d])^2
// PASS 2: PASS 0 plus, in the same sweep, S2'[k][d] = sum_n w_k(n) (y[n][d] - c[k][d])^2 about a
//         SHIFT c (the previous iteration's mean, or the first row of y when there is none):
//         the finalize turns it into the variance about the new mean without a second pass over
//         the embedding; with c within the data's spread there is no cancellation to speak of.
// part layout: [b][chunk][k][E+1]  (slot E = S0); PASS 2 writes S2' to part2 (same layout)
constexpr int kFitThreads = 1024;  // 16 waves: one workgroup per CU
constexpr int kFitUnroll = 16;     // x 16 loads in flight per thread

template <int K, typename TS, int PASS>
__global__ void __launch_bounds__(kFitThreads)
    embed_fit_kernel(const TS* yr, int64_t N, int E, int S, int C, int64_t L, const double* aff,
                     int64_t Tin, const double* sal, const double* mean, double* part,
                     double* part2, int shift_first_row, const double* rowscale) {
  extern __shared__ double sm[];
  double* red = sm;               // [S][K][E]
  double* red0 = sm + S * K * E;  // [S][K]
  double* red2 = red0 + S * K;    // [S][K][E]  (PASS 2)
  const int64_t b = blockIdx.y;
  const int c = blockIdx.x;
  const int tid = threadIdx.x;
  const int s = tid / E;
  const int d = tid - s * E;
  const bool active = s < S;
  const int64_t n0 = (int64_t)c * L;
  const int64_t n1 = (n0 + L < N) ? (n0 + L) : N;
  double acc[K], acc0[K], acc2[K], mu[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    acc[k] = 0.0;
    acc0[k] = 0.0;
    acc2[k] = 0.0;
    mu[k] = 0.0;
    if (PASS == 1 && active) mu[k] = mean[((size_t)b * K + k) * E + d];
    if (PASS == 2 && active)
      mu[k] = shift_first_row ? (double)yr[(size_t)b * N * E + d] : mean[((size_t)b * K + k) * E + d];
  }
  // Per trip the workgroup covers U * S samples.  Their class weights affiliation * saliency
  // (vmfmm.py:167) are staged through LDS first: every one of the E threads of a sample needs
  // the same K values, and as global loads those were E-fold redundant requests on the L1
  // (K + 1 of the K + 2 loads per element).
  constexpr int U = kFitUnroll;  // loads in flight per thread: the sweep is latency-bound (1 workgroup per CU)
  double* wst = red2 + (size_t)S * K * E;  // [U * S][K]
  const TS* base = yr + (size_t)b * N * E;
  const int SU = U * S;
  // rowscale (vMF mixture): the rows are used as y_n * rowscale[n] (unit norm, vmfmm.py:76-78)
  // without a normalised copy of the embedding -- the scale rides on the staged weights of the
  // first moments; the weight sums S0 need the plain weights and are collected by the staging
  // loop itself (a conditional second read inside the accumulation loop cost 6x the kernel)
  double s0acc[K];  // rowscale: this thread's share of S0[k], collected while staging
#pragma unroll
  for (int k = 0; k < K; ++k) s0acc[k] = 0.0;
  for (int64_t nb = n0; nb < n1; nb += SU) {
    for (int i = tid; i < SU * K; i += kFitThreads) {
      const int k = i / SU, smp = i - k * SU;  // consecutive lanes = consecutive samples
      const int64_t nn = nb + smp;
      double wv = 0.0, sc = 1.0;
      if (nn < n1) {
        wv = aff[aff_index(b, k, nn, K, N, Tin)];
        if (sal) wv *= sal[(size_t)b * N + nn];
        if (rowscale) sc = rowscale[(size_t)b * N + nn];
      }
      wst[smp * K + k] = wv * sc;
      if (rowscale) {
#pragma unroll
        for (int kk = 0; kk < K; ++kk) s0acc[kk] += (kk == k) ? wv : 0.0;
      }
    }
    __syncthreads();
    if (active) {
      TS yv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t nn = nb + s + (int64_t)u * S;
        const int64_t nc = (nn < n1) ? nn : nb;  // staged weight is 0 there
        yv[u] = base[(size_t)nc * E + d];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const double wk = wst[(s + u * S) * K + k];
          if (PASS == 0 || PASS == 2) {
            acc[k] = fma(wk, (double)yv[u], acc[k]);
            acc0[k] += wk;  // (with rowscale: scaled, unused -- S0 comes from the staging loop)
          }
          if (PASS == 1) {
            const double df = (double)yv[u] - mu[k];
            acc[k] = fma(wk * df, df, acc[k]);
          }
          if (PASS == 2) {
            const double df = (double)yv[u] - mu[k];
            acc2[k] = fma(wk * df, df, acc2[k]);
          }
        }
      }
    }
    __syncthreads();
  }
  if (active) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      red[(s * K + k) * E + d] = acc[k];
      if (PASS != 1 && d == 0) red0[s * K + k] = acc0[k];
      if (PASS == 2) red2[(s * K + k) * E + d] = acc2[k];
    }
  }
  __syncthreads();
  double* dst = part + ((size_t)b * C + c) * K * (E + 1);
  for (int i = tid; i < K * E; i += kFitThreads) {
    const int k = i / E, dd = i - k * E;
    double t = 0.0;
    for (int ss = 0; ss < S; ++ss) t += red[(ss * K + k) * E + dd];
    dst[k * (E + 1) + dd] = t;
    if (PASS == 2) {
      double t2 = 0.0;
      for (int ss = 0; ss < S; ++ss) t2 += red2[(ss * K + k) * E + dd];
      part2[((size_t)b * C + c) * K * (E + 1) + k * (E + 1) + dd] = t2;
    }
  }
  if (rowscale) {  // S0 from the plain weights: wave sums, then the 16 waves in order
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const double t = wave_sum(s0acc[k]);
      if ((tid & (kWave - 1)) == 0) wst[(tid / kWave) * K + k] = t;
    }
    __syncthreads();
    if (tid < K) {
      double t = 0.0;
      for (int w = 0; w < kFitThreads / kWave; ++w) t += wst[w * K + tid];
      dst[tid * (E + 1) + E] = t;
    }
  } else if (PASS != 1 && tid < K) {
    double t = 0.0;
    for (int ss = 0; ss < S; ++ss) t += red0[ss * K + tid];
    dst[tid * (E + 1) + E] = t;
  }
}

// ---------------------------------------------------------------- fused E + M sweep (vMF mixture)
// Round 3: ONE pass over the embedding per EM iteration.  The E-step of iteration i and the
// M-step partial sums of iteration i consume the same rows, so a workgroup stages a tile of R
// consecutive rows of the caller's row-major (N, E) array in LDS (one contiguous, fully coalesced
// block of R * E elements; LDS row stride E | 1: odd, conflict-free both ways) and uses it twice:
//   E  thread = row: |y_n|, the K dot products with the class means, softmax -> w_k(n) = gamma_k(n)
//      * saliency(n) and 1 / |y_n| into LDS (the first iteration takes gamma from the caller's
//      initialisation instead; the last call may also write the affiliations)
//   M  thread = (row slot s, dimension d): S1[k][d] += w_k(n) y[n][d] / |y_n|, S0[k] += w_k(n)
// then the slots are folded through LDS and the workgroup writes ONE chunk partial in exactly the
// layout of embed_fit_kernel -- embed_finalize_kernel is reused unchanged.  No transposed copy, no
// row-scale array, no (B, K, N) affiliation round trip through HBM between the two steps: 41 MB
// instead of 2 x 41 + 2 x 6 MB per iteration for N = 256 500, E = 40 float32.
constexpr int kFusedThreads = 256;

template <int K, typename TS, bool VEC>
__global__ void __launch_bounds__(kFusedThreads)
    vmf_em_kernel(const TS* __restrict__ y, int64_t N, int E, int R, int C, int64_t L,
                  const double* __restrict__ gamma, const double* __restrict__ mean,
                  const double* __restrict__ prec, const double* __restrict__ offset,
                  const double* __restrict__ weight, const double* __restrict__ sal,
                  double* __restrict__ part, double* __restrict__ out_aff) {
  extern __shared__ __attribute__((aligned(16))) char smraw[];
  constexpr int KW = (K + 1) & ~1;  // class weights of a row padded to whole 16-byte reads
  const int ES = E | 1;             // LDS row stride (odd)
  const int S = kFusedThreads / E;  // row slots of the M part (E <= 256: S >= 1)
  double* affw = reinterpret_cast<double*>(smraw);  // [R][KW]  gamma * saliency / |y_n|
  double* red = affw + (size_t)R * KW;              // [S][K][E]
  double* red0 = red + (size_t)S * K * E;           // [waves][K]
  TS* tile = reinterpret_cast<TS*>(red0 + (kFusedThreads / kWave) * K);  // [R][ES]
  const int64_t b = blockIdx.y;
  const int c = blockIdx.x;
  const int tid = threadIdx.x;
  // class means / constants of the E part are wave-uniform: read through the scalar cache
  // (s_load, SGPR operands of the FMAs) straight from the model arrays -- no LDS copy, no
  // broadcast reads (the LDS pipe is what bounds this kernel: tile in, tile out twice)
  const double* mu = mean ? mean + (size_t)b * K * E : nullptr;
  const int s = tid / E, d = tid - s * E;
  const bool active = s < S;
  double acc[K], s0[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    acc[k] = 0.0;
    s0[k] = 0.0;
  }
  const int64_t n0 = (int64_t)c * L;
  const int64_t n1 = (n0 + L < N) ? (n0 + L) : N;
  // Tile load: the R rows of a tile are ONE contiguous block of rows * E elements.  With VEC
  // (E a multiple of the 16-byte vector: the usual case) a thread fetches kVU 16-byte vectors --
  // the whole tile is in flight at once -- and the NEXT tile's vectors are requested before the
  // current tile is consumed (registers), so the memory latency of a tile hides behind the E and M
  // parts of its predecessor.  (Eight 4-byte loads per thread and batch were five dependent
  // memory round trips per tile: 24 us per sweep)
  constexpr int VW = 16 / (int)sizeof(TS);
  typedef TS VecT __attribute__((ext_vector_type(VW)));
  constexpr int kVU = 12;  // vectors per thread: covers R * E <= 12 * 256 * VW elements
  const bool vec = VEC && (R * E <= kVU * kFusedThreads * VW);
  VecT pre[kVU];
  auto request = [&](int64_t nb2) {  // issue the loads of the tile that starts at row nb2
    const int rows2 = (int)((n1 - nb2 < R) ? (n1 - nb2) : R);
    const int nv = rows2 * E / VW;
    const VecT* base = reinterpret_cast<const VecT*>(y + ((size_t)b * N + nb2) * E);
#pragma unroll
    for (int u = 0; u < kVU; ++u) {
      const int i = tid + u * kFusedThreads;
      pre[u] = base[i < nv ? i : (nv > 0 ? nv - 1 : 0)];  // clamped, masked at the LDS store
    }
  };
  if (vec && n0 < n1) request(n0);
  for (int64_t nb = n0; nb < n1; nb += R) {
    const int rows = (int)((n1 - nb < R) ? (n1 - nb) : R);
    __syncthreads();  // previous tile fully consumed
    if (vec) {
      const int nv = rows * E / VW;
#pragma unroll
      for (int u = 0; u < kVU; ++u) {
        const int i = tid + u * kFusedThreads;
        if (i < nv) {
          const int e0 = i * VW;
          const int r = e0 / E;
          TS* dst = tile + (size_t)r * ES + (e0 - r * E);
#pragma unroll
          for (int x = 0; x < VW; ++x) dst[x] = pre[u][x];
        }
      }
      if (nb + R < n1) request(nb + R);
    } else {  // generic: 4- / 8-byte loads, eight in flight per thread
      const TS* base = y + ((size_t)b * N + nb) * E;
      const int total = rows * E;
      constexpr int U = 8;
      for (int i0 = tid; i0 < total; i0 += U * kFusedThreads) {
        TS raw[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int i = i0 + u * kFusedThreads;
          raw[u] = base[i < total ? i : total - 1];  // clamped, masked below (no guarded loads)
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int i = i0 + u * kFusedThreads;
          if (i < total) {
            const int r = i / E;
            tile[(size_t)r * ES + (i - r * E)] = raw[u];
          }
        }
      }
    }
    __syncthreads();
    // ---- E part: thread = row
    if (tid < R) {
      double w[K];
#pragma unroll
      for (int k = 0; k < K; ++k) w[k] = 0.0;
      if (tid < rows) {
        const TS* row = tile + (size_t)tid * ES;
        const int64_t n = nb + tid;
        double n2 = 0.0, dot[K];
#pragma unroll
        for (int k = 0; k < K; ++k) dot[k] = 0.0;
        if (mu) {
          for (int e = 0; e < E; ++e) {
            const double v = (double)row[e];
            n2 = fma(v, v, n2);
#pragma unroll
            for (int k = 0; k < K; ++k) dot[k] = fma(v, mu[k * E + e], dot[k]);
          }
        } else {
          for (int e = 0; e < E; ++e) {
            const double v = (double)row[e];
            n2 = fma(v, v, n2);
          }
        }
        const double inv = 1.0 / fmax(sqrt(n2), kTiny);  // unit rows, vmfmm.py:76-78
        double g[K];
        if (gamma) {
#pragma unroll
          for (int k = 0; k < K; ++k) g[k] = gamma[((size_t)b * K + k) * N + n];
        } else {
          double lp[K], mx = -1.79e308;
#pragma unroll
          for (int k = 0; k < K; ++k) {
            lp[k] = fma(prec[b * K + k], dot[k] * inv, offset[b * K + k]);  // von_mises_fisher.py:71-77
            mx = fmax(mx, lp[k]);
          }
          double den = 0.0;
#pragma unroll
          for (int k = 0; k < K; ++k) {
            g[k] = exp(lp[k] - mx) * (weight ? weight[b * K + k] : 1.0);  // mixture_model_utils.py:30-47
            den += g[k];
          }
          den = fmax(den, kTiny);
#pragma unroll
          for (int k = 0; k < K; ++k) g[k] /= den;
        }
        if (out_aff) {
#pragma unroll
          for (int k = 0; k < K; ++k) out_aff[((size_t)b * K + k) * N + n] = g[k];
        }
        const double sv = sal ? sal[(size_t)b * N + n] : 1.0;  // vmfmm.py:167
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const double wk = g[k] * sv;
          s0[k] += wk;          // S0 takes the plain weights
          w[k] = wk * inv;      // the first moments take unit rows: the scale rides on the weight
        }
      }
#pragma unroll
      for (int k = 0; k < KW; ++k) affw[tid * KW + k] = (k < K) ? w[k < K ? k : 0] : 0.0;
    }
    __syncthreads();
    // ---- M part: thread = (slot, dimension)
    if (active) {
      for (int r = s; r < rows; r += S) {
        const double v = (double)tile[(size_t)r * ES + d];
        double wk[KW];
#pragma unroll
        for (int k = 0; k < KW; k += 2) {
          const double2 p2 = *reinterpret_cast<const double2*>(affw + r * KW + k);
          wk[k] = p2.x;
          wk[k + 1] = p2.y;
        }
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] = fma(wk[k], v, acc[k]);
      }
    }
  }
  __syncthreads();
  if (active) {
#pragma unroll
    for (int k = 0; k < K; ++k) red[((size_t)s * K + k) * E + d] = acc[k];
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const double t = wave_sum(s0[k]);
    if ((tid & (kWave - 1)) == 0) red0[(tid / kWave) * K + k] = t;
  }
  __syncthreads();
  double* dst = part + ((size_t)b * C + c) * K * (E + 1);
  for (int i = tid; i < K * E; i += kFusedThreads) {
    const int k = i / E, dd = i - k * E;
    double t = 0.0;
    for (int ss = 0; ss < S; ++ss) t += red[((size_t)ss * K + k) * E + dd];
    dst[k * (E + 1) + dd] = t;
  }
  if (tid < K) {
    double t = 0.0;
    for (int w = 0; w < kFusedThreads / kWave; ++w) t += red0[w * K + tid];
    dst[tid * (E + 1) + E] = t;
  }
}

// ---------------------------------------------------------------- persistent vMF mixture (many small mixtures)
// Round 4 (BASELINE configs[3], vMF leg: one mixture per frequency bin, B = 257 mixtures of N = 800
// points, E = 12).  A mixture that fits ONE workgroup's LDS needs no partials, no finalize kernel
// and no relaunch: the workgroup stages its rows once and runs the WHOLE EM loop -- E part
// (thread = row), M part (thread = (row slot, dimension)), slot reduction, model (mean, the
// concentration of Banerjee 2005 eq. 4.4, the log-Bessel normaliser, the mixture weights) -- with
// the model in LDS between the iterations (vmfmm.py:124-172, von_mises_fisher.py:28-144).  The
// two-launches-per-iteration path took 26.8 us per iteration for that shape.
template <int K, typename TS>
__global__ void __launch_bounds__(kFusedThreads)
    vmf_bin_em_kernel(const TS* __restrict__ y, int64_t N, int E, int iterations,
                      const double* __restrict__ gamma, const double* __restrict__ sal,
                      const double* __restrict__ in_mean, const double* __restrict__ in_conc,
                      const double* __restrict__ in_weight, double cmin, double cmax,
                      int weight_mode, double* __restrict__ out_mean, double* __restrict__ out_conc,
                      double* __restrict__ out_weight, double* __restrict__ out_aff) {
  extern __shared__ __attribute__((aligned(16))) char smraw[];
  constexpr int KW = (K + 1) & ~1;
  constexpr int kWaves = kFusedThreads / kWave;
  const int ES = E | 1;
  const int S = kFusedThreads / E;
  double* affw = reinterpret_cast<double*>(smraw);          // [N][KW]  gamma * saliency / |y_n|
  double* red = affw + (size_t)N * KW;                      // [S][K][E]
  double* red0 = red + (size_t)S * K * E;                   // [waves][K]
  double* mu = red0 + kWaves * K;                           // [K][E]
  double* tot = mu + (size_t)K * E;                         // [K][E + 1]
  double* prec = tot + (size_t)K * (E + 1);                 // [K]
  double* off = prec + K;                                   // [K]
  double* wgt = off + K;                                    // [K]
  TS* tile = reinterpret_cast<TS*>(wgt + K);                // [N][ES]
  const int64_t b = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1), wave = tid / kWave;
  const TS* base = y + (size_t)b * N * E;
  const int total = (int)N * E;
  {  // stage the rows: unconditional loads at clamped indices, 8 in flight per thread
    constexpr int U = 8;
    for (int i0 = tid; i0 < total; i0 += U * kFusedThreads) {
      TS raw[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * kFusedThreads;
        raw[u] = base[i < total ? i : total - 1];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * kFusedThreads;
        if (i < total) {
          const int r = i / E;
          tile[(size_t)r * ES + (i - r * E)] = raw[u];
        }
      }
    }
  }
  if (!gamma) {  // predict only (iterations == 0): the model comes from the caller
    for (int i = tid; i < K * E; i += kFusedThreads) mu[i] = in_mean[(size_t)b * K * E + i];
    for (int k = wave; k < K; k += kWaves) {
      const double conc = in_conc[b * K + k];
      const double o = -(0.5 * E * kLn2Pi + wave_log_bessel_over_power(0.5 * E - 1.0, conc, lane));
      if (lane == 0) {
        prec[k] = conc;
        off[k] = o;
        wgt[k] = in_weight[b * K + k];
      }
    }
  }
  __syncthreads();
  const int s = tid / E, d = tid - s * E;
  const bool active = s < S;
  // E part of one sweep: posteriors of the model in LDS (or the caller's initialisation), the
  // M-step weights into affw, the plain weight sums per class; FINAL: affiliations to out_aff
  auto e_part = [&](bool from_gamma, bool final_pass) {
    double s0[K];
#pragma unroll
    for (int k = 0; k < K; ++k) s0[k] = 0.0;
    for (int n = tid; n < (int)N; n += kFusedThreads) {
      const TS* row = tile + (size_t)n * ES;
      double n2 = 0.0, dot[K];
#pragma unroll
      for (int k = 0; k < K; ++k) dot[k] = 0.0;
      if (from_gamma) {
        for (int e = 0; e < E; ++e) {
          const double v = (double)row[e];
          n2 = fma(v, v, n2);
        }
      } else {
        for (int e = 0; e < E; ++e) {
          const double v = (double)row[e];
          n2 = fma(v, v, n2);
#pragma unroll
          for (int k = 0; k < K; ++k) dot[k] = fma(v, mu[k * E + e], dot[k]);
        }
      }
      const double inv = 1.0 / fmax(sqrt(n2), kTiny);  // unit rows, vmfmm.py:76-78
      double g[K];
      if (from_gamma) {
#pragma unroll
        for (int k = 0; k < K; ++k) g[k] = gamma[((size_t)b * K + k) * N + n];
      } else {
        double lp[K], mx = -1.79e308;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          lp[k] = fma(prec[k], dot[k] * inv, off[k]);  // von_mises_fisher.py:71-77
          mx = fmax(mx, lp[k]);
        }
        double den = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          g[k] = exp(lp[k] - mx) * wgt[k];  // mixture_model_utils.py:30-47
          den += g[k];
        }
        den = fmax(den, kTiny);
#pragma unroll
        for (int k = 0; k < K; ++k) g[k] /= den;
      }
      if (final_pass) {
#pragma unroll
        for (int k = 0; k < K; ++k) out_aff[((size_t)b * K + k) * N + n] = g[k];
      } else {
        const double sv = sal ? sal[(size_t)b * N + n] : 1.0;  // vmfmm.py:167
#pragma unroll
        for (int k = 0; k < KW; ++k) {
          const double wk = (k < K) ? g[k < K ? k : 0] * sv : 0.0;
          if (k < K) s0[k] += wk;
          affw[(size_t)n * KW + k] = wk * inv;
        }
      }
    }
    if (!final_pass) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const double t = wave_sum(s0[k]);
        if (lane == 0) red0[wave * K + k] = t;
      }
    }
  };
  for (int it = 0; it < iterations; ++it) {
    e_part(it == 0 && gamma != nullptr, false);
    __syncthreads();
    // ---- M part: thread = (slot, dimension) over all rows
    if (active) {
      double acc[K];
#pragma unroll
      for (int k = 0; k < K; ++k) acc[k] = 0.0;
      for (int r = s; r < (int)N; r += S) {
        const double v = (double)tile[(size_t)r * ES + d];
        double wk[KW];
#pragma unroll
        for (int k = 0; k < KW; k += 2) {
          const double2 p2 = *reinterpret_cast<const double2*>(affw + (size_t)r * KW + k);
          wk[k] = p2.x;
          wk[k + 1] = p2.y;
        }
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] = fma(wk[k], v, acc[k]);
      }
#pragma unroll
      for (int k = 0; k < K; ++k) red[((size_t)s * K + k) * E + d] = acc[k];
    }
    __syncthreads();
    for (int i = tid; i < K * E; i += kFusedThreads) {
      const int k = i / E, dd = i - k * E;
      double t = 0.0;
      for (int ss = 0; ss < S; ++ss) t += red[((size_t)ss * K + k) * E + dd];
      tot[k * (E + 1) + dd] = t;
    }
    if (tid < K) {
      double t = 0.0;
      for (int w = 0; w < kWaves; ++w) t += red0[w * K + tid];
      tot[tid * (E + 1) + E] = t;
    }
    __syncthreads();
    // ---- model: one wavefront per class (von_mises_fisher.py:122-144)
    for (int k = wave; k < K; k += kWaves) {
      const double* row = tot + k * (E + 1);
      const double s0k = row[E];
      double n2 = 0.0;
      for (int dd = lane; dd < E; dd += kWave) n2 = fma(row[dd], row[dd], n2);
      n2 = wave_sum(n2);
      const double norm = sqrt(n2);
      const double rn = 1.0 / fmax(norm, kTiny);                            // eq. 2.4
      for (int dd = lane; dd < E; dd += kWave) mu[k * E + dd] = row[dd] * rn;
      const double rbar = norm / s0k;                                        // eq. 2.5
      double conc = (rbar * E - rbar * rbar * rbar) / (1.0 - rbar * rbar);  // eq. 4.4
      conc = conc < cmin ? cmin : (conc > cmax ? cmax : conc);              // NaN stays NaN
      const double o = -(0.5 * E * kLn2Pi + wave_log_bessel_over_power(0.5 * E - 1.0, conc, lane));
      if (lane == 0) {
        prec[k] = conc;
        off[k] = o;
      }
    }
    if (tid == 0) {
      if (weight_mode == 1) {
        for (int k = 0; k < K; ++k) wgt[k] = 1.0 / K;
      } else {  // estimate_mixture_weight with saliency: L1 unit norm, eps 'where' 1e-10
        double t = 0.0;
        for (int k = 0; k < K; ++k) t += fabs(tot[k * (E + 1) + E]);
        if (t == 0.0) t = 1e-10;
        for (int k = 0; k < K; ++k) wgt[k] = tot[k * (E + 1) + E] / t;
      }
    }
    __syncthreads();
  }
  if (iterations > 0) {
    for (int i = tid; i < K * E; i += kFusedThreads) out_mean[(size_t)b * K * E + i] = mu[i];
    if (tid < K) {
      out_conc[b * K + tid] = prec[tid];
      out_weight[b * K + tid] = wgt[tid];
    }
  }
  if (out_aff) e_part(false, true);
}

// ---------------------------------------------------------------- joint models: fused sweep
// Round 4 (the "rotated" joint loop of pbbss_joint_fit): ONE pass over the
// embedding per EM iteration for GCACGMM (spherical) / VMFCACGMM.  The tile machinery is the one
// of vmf_em_kernel; the E part of a row n = (bin f, frame t) additionally takes the spatial half
// of the posterior from the quadratic forms Q[f, k, t] and ln det B_fk that the spatial kernel
// (cacgmm_em.hpp: run_joint_ms) left behind:
//   log p_k = spatial_weight * (-D ln Q_fkt - ln det B_fk) + spectral_weight * spectral_k(e_n)
//   gamma   = softmax_k(log p) * w      (gcacgmm.py:66-117, vmfcacgmm.py:57-97,
//                                        mixture_model_utils.py:30-53: clip to [eps, 1 - eps])
// gamma goes to G (F, K, T) for the spatial M-step; gamma * saliency are the weights of the
// spectral M-step sums taken from the SAME tile: S1 = sum w e, S0 = sum w and, for the spherical
// Gaussian, the second moment about the shift c = the mean the E part has just used (the finalize
// corrects it to the new mean).  Only the TOTAL over the dimensions is needed ('spherical'), and
// the E part has |e_n - c_k|^2 of every row in hand for the log-pdf: S2 = sum_n w_nk |e_n - c_k|^2
// costs one fma per row and class (it was three VALU operations per row, class and DIMENSION in
// the M part: 26.7 -> 22.4 us per sweep... see profiles/r04_m_gauss_sweep.txt).  The vMF half normalises the rows
// in the E part only -- the reference's M-step takes the embedding as given (vmfcacgmm.py:286,
// VonMisesFisherTrainer._fit).
template <int KIND, int K, typename TS, bool VEC>
__global__ void __launch_bounds__(kFusedThreads)
    joint_sweep_kernel(const TS* __restrict__ y, int64_t N, int E, int R, int C, int64_t L, int T,
                       int D, const double* __restrict__ Q, const double* __restrict__ lndet,
                       const double* __restrict__ weight, int64_t wb, int64_t wk, int64_t wt,
                       const double* __restrict__ mean, const double* __restrict__ prec,
                       const double* __restrict__ offset, double spatial_weight,
                       double spectral_weight, const double* __restrict__ sal, double eps,
                       double* __restrict__ G, double* __restrict__ part,
                       double* __restrict__ part2) {
  extern __shared__ __attribute__((aligned(16))) char smraw[];
  constexpr bool GAUSS = KIND != PBBSS_EMBED_VMF;
  constexpr int KW = (K + 1) & ~1;
  const int ES = E | 1;
  const int S = kFusedThreads / E;
  double* affw = reinterpret_cast<double*>(smraw);   // [R][KW]  gamma * saliency
  double* red0 = affw + (size_t)R * KW;              // [waves][K]
  TS* tile = reinterpret_cast<TS*>(red0 + (kFusedThreads / kWave) * K);  // [R][ES]
  // the slot-reduction buffers alias the tile: they are written after the last tile has been
  // consumed (barrier below).  11.5 KB less LDS per workgroup: three workgroups per CU.
  double* red = reinterpret_cast<double*>(tile);     // [S][K][E]
  double* red2 = red + (size_t)S * K * E;            // [waves][K] second moments (Gaussian)
  const int c = blockIdx.x;
  const int tid = threadIdx.x;
  const int s = tid / E, d = tid - s * E;
  const bool active = s < S;
  double acc[K], s0[K], s2[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    acc[k] = 0.0;
    s0[k] = 0.0;
    s2[k] = 0.0;  // Gaussian: sum w |e - c|^2 about the shift c = the mean of this sweep
  }
  const int64_t n0 = (int64_t)c * L;
  const int64_t n1 = (n0 + L < N) ? (n0 + L) : N;
  constexpr int VW = 16 / (int)sizeof(TS);
  typedef TS VecT __attribute__((ext_vector_type(VW)));
  constexpr int kVU = 12;
  const bool vec = VEC && (R * E <= kVU * kFusedThreads * VW);
  VecT pre[kVU];
  auto request = [&](int64_t nb2) {
    const int rows2 = (int)((n1 - nb2 < R) ? (n1 - nb2) : R);
    const int nv = rows2 * E / VW;
    const VecT* base = reinterpret_cast<const VecT*>(y + (size_t)nb2 * E);
#pragma unroll
    for (int u = 0; u < kVU; ++u) {
      const int i = tid + u * kFusedThreads;
      pre[u] = base[i < nv ? i : (nv > 0 ? nv - 1 : 0)];
    }
  };
  if (vec && n0 < n1) request(n0);
  for (int64_t nb = n0; nb < n1; nb += R) {
    const int rows = (int)((n1 - nb < R) ? (n1 - nb) : R);
    // the row's spatial inputs are requested before the tile lands in LDS (independent loads)
    double qv[K], ld[K], wv[K], sv = 1.0;
    int64_t gidx = 0;
    const bool mine = tid < rows;
    {
      const int64_t n = nb + (mine ? tid : 0);
      const int64_t f = n / T;
      const int t = (int)(n - f * T);
      gidx = (f * K) * (int64_t)T + t;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        qv[k] = Q[gidx + (int64_t)k * T];
        ld[k] = lndet[f * K + k];
        wv[k] = weight ? weight[f * wb + k * wk + (int64_t)t * wt] : 1.0;
      }
      if (sal) sv = sal[n];
    }
    __syncthreads();  // previous tile fully consumed
    if (vec) {
      const int nv = rows * E / VW;
#pragma unroll
      for (int u = 0; u < kVU; ++u) {
        const int i = tid + u * kFusedThreads;
        if (i < nv) {
          const int e0 = i * VW;
          const int r = e0 / E;
          TS* dst = tile + (size_t)r * ES + (e0 - r * E);
#pragma unroll
          for (int x = 0; x < VW; ++x) dst[x] = pre[u][x];
        }
      }
      if (nb + R < n1) request(nb + R);
    } else {
      const TS* base = y + (size_t)nb * E;
      const int total = rows * E;
      constexpr int U = 8;
      for (int i0 = tid; i0 < total; i0 += U * kFusedThreads) {
        TS raw[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int i = i0 + u * kFusedThreads;
          raw[u] = base[i < total ? i : total - 1];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int i = i0 + u * kFusedThreads;
          if (i < total) {
            const int r = i / E;
            tile[(size_t)r * ES + (i - r * E)] = raw[u];
          }
        }
      }
    }
    __syncthreads();
    // ---- E part: thread = row
    if (tid < R) {
      double w[K];
#pragma unroll
      for (int k = 0; k < K; ++k) w[k] = 0.0;
      if (mine) {
        const TS* row = tile + (size_t)tid * ES;
        double n2 = 0.0, a[K], d2[K];
#pragma unroll
        for (int k = 0; k < K; ++k) a[k] = d2[k] = 0.0;
        if (GAUSS) {
          for (int e = 0; e < E; ++e) {
            const double v = (double)row[e];
#pragma unroll
            for (int k = 0; k < K; ++k) {
              const double u = v - mean[k * E + e];  // gaussian.py:127-131; the common
              a[k] = fma(u, u, a[k]);                // precision factor is applied below
            }
          }
#pragma unroll
          for (int k = 0; k < K; ++k) {
            d2[k] = a[k];  // |e - mean_k|^2: also this row's share of the second moment
            a[k] *= prec[k] * prec[k];
          }
        } else {
          for (int e = 0; e < E; ++e) {
            const double v = (double)row[e];
            n2 = fma(v, v, n2);
#pragma unroll
            for (int k = 0; k < K; ++k) a[k] = fma(v, mean[k * E + e], a[k]);
          }
        }
        // The transcendental part of a row (three logarithms, three exponentials, a division, a
        // square root) was ~300 of the ~1 450 VALU instructions a lane spends per tile with the
        // library calls; the mantissa / exponent forms of pbbss_dev.hpp (~2 ulp) are a third of that.
        // Q >= tiny (normal) by construction; NaN passes through, +Inf takes the branch.
        const double inv = GAUSS ? 0.0
                                 : ((n2 > 1e-280 && n2 < 1e280) ? fast_rsqrt(n2)
                                                                 : 1.0 / fmax(sqrt(n2), kTiny));
        double lp[K], mx = -1.79e308;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const double spec = GAUSS ? offset[k] - 0.5 * a[k]                // gaussian.py:132-136
                                    : fma(prec[k], a[k] * inv, offset[k]);  // von_mises_fisher.py:71-77
          const double lq = (qv[k] >= kTiny && qv[k] < 1.79e308) ? log_pos(qv[k]) : log(qv[k]);
          const double spat = -(double)D * lq - ld[k];                      // cacg.py:200-201
          lp[k] = spatial_weight * spat + spectral_weight * spec;
          mx = fmax(mx, lp[k]);
        }
        double g[K], den = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          g[k] = exp_nonpos(lp[k] - mx) * wv[k];
          den += g[k];
        }
        // 1 / max(den, tiny) (mixture_model_utils.py:43-47) by a Newton-refined reciprocal; den - den
        // carries a NaN / Inf denominator into the posterior exactly as the division did
        const double rden = fast_rcp(fmax(den, kTiny)) + (den - den);
#pragma unroll
        for (int k = 0; k < K; ++k) {
          double gam = g[k] * rden;
          if (eps != 0.0) gam = fmin(fmax(gam, eps), 1.0 - eps);
          G[gidx + (int64_t)k * T] = gam;
          const double wk2 = gam * sv;
          s0[k] += wk2;
          if (GAUSS) s2[k] = fma(wk2, d2[k], s2[k]);
          w[k] = wk2;
        }
      }
#pragma unroll
      for (int k = 0; k < KW; ++k) affw[tid * KW + k] = (k < K) ? w[k < K ? k : 0] : 0.0;
    }
    __syncthreads();
    // ---- M part: thread = (slot, dimension)
    if (active) {
      for (int r = s; r < rows; r += S) {
        const double v = (double)tile[(size_t)r * ES + d];
        double wk3[KW];
#pragma unroll
        for (int k = 0; k < KW; k += 2) {
          const double2 p2 = *reinterpret_cast<const double2*>(affw + r * KW + k);
          wk3[k] = p2.x;
          wk3[k + 1] = p2.y;
        }
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] = fma(wk3[k], v, acc[k]);
      }
    }
  }
  __syncthreads();
  if (active) {
#pragma unroll
    for (int k = 0; k < K; ++k) red[((size_t)s * K + k) * E + d] = acc[k];
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const double t = wave_sum(s0[k]);
    if ((tid & (kWave - 1)) == 0) red0[(tid / kWave) * K + k] = t;
    if (GAUSS) {
      const double t2 = wave_sum(s2[k]);
      if ((tid & (kWave - 1)) == 0) red2[(tid / kWave) * K + k] = t2;
    }
  }
  __syncthreads();
  double* dst = part + (size_t)c * K * (E + 1);
  for (int i = tid; i < K * E; i += kFusedThreads) {
    const int k = i / E, dd = i - k * E;
    double t = 0.0;
    for (int ss = 0; ss < S; ++ss) t += red[((size_t)ss * K + k) * E + dd];
    dst[k * (E + 1) + dd] = t;
    if (GAUSS) {
      // the finalize adds the per-dimension second moments up anyway ('spherical'): the whole
      // moment travels in dimension 0
      double t2 = 0.0;
      if (dd == 0)
        for (int w = 0; w < kFusedThreads / kWave; ++w) t2 += red2[w * K + k];
      part2[(size_t)c * K * (E + 1) + k * (E + 1) + dd] = t2;
    }
  }
  if (tid < K) {
    double t = 0.0;
    for (int w = 0; w < kFusedThreads / kWave; ++w) t += red0[w * K + tid];
    dst[tid * (E + 1) + E] = t;
  }
}

// ---------------------------------------------------------------- joint models: wave-private sweep
// Round 6.  joint_sweep_kernel above walks its chunk in tiles of 256 rows with three workgroup
// barriers per tile, and a chunk is two tiles: every workgroup of the launch loads, computes the E
// part and computes the M part in lockstep with all the others -- the memory system and the vector
// units take turns (54 MB in 22.4 us with ~9 us of issue, profiles/r04_m_joint_pmc.txt).  Here a
// WAVEFRONT owns a sub-tile of 64 rows (lane = row in the E part, lane = dimension in the M part)
// in its own LDS region: no workgroup barrier inside the sweep, the four wavefronts of a
// workgroup drift apart and the loads of one overlap the arithmetic of another; the next sub-tile
// of the wavefront is requested into registers before the current one is consumed.  Chunk
// geometry (C chunks of L rows, one partial per chunk) and the partials' layout are those of
// joint_sweep_kernel, so the finalize (helper blocks of the spatial kernel, embed_dev.hpp) does
// not change.  float32 embeddings, E % 4 == 0, E <= 4 NU <= 64; anything else takes the kernel
// above.
__device__ __forceinline__ void sweep_lds_order() { __asm__ volatile("" ::: "memory"); }

template <int KIND, int K, int NU>
__global__ void __launch_bounds__(kFusedThreads)
    joint_sweep_wave_kernel(const float* __restrict__ y, int64_t N, int E, int64_t L, int T, int D,
                            const double* __restrict__ Q, const double* __restrict__ lndet,
                            const double* __restrict__ weight, int64_t wb, int64_t wk, int64_t wt,
                            const double* __restrict__ mean, const double* __restrict__ prec,
                            const double* __restrict__ offset, double spatial_weight,
                            double spectral_weight, const double* __restrict__ sal, double eps,
                            double* __restrict__ G, double* __restrict__ part,
                            double* __restrict__ part2) {
  extern __shared__ __attribute__((aligned(16))) char smraw[];
  constexpr bool GAUSS = KIND != PBBSS_EMBED_VMF;
  constexpr int KW = (K + 1) & ~1;
  constexpr int NWAVE = kFusedThreads / kWave;
  const int ES = E | 1;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & (kWave - 1);
  const size_t wbytes = (size_t)kWave * ES * 4 + (size_t)kWave * KW * 8;  // multiple of 16
  float* tile = reinterpret_cast<float*>(smraw + wave * wbytes);         // [64][ES]
  double* gam = reinterpret_cast<double*>(tile + (size_t)kWave * ES);    // [64][KW] gamma * saliency
  double* red = reinterpret_cast<double*>(smraw + NWAVE * wbytes);       // [NWAVE][K][E]
  double* red0 = red + (size_t)NWAVE * K * E;                            // [NWAVE][K]
  double* red2 = red0 + NWAVE * K;                                       // [NWAVE][K]
  const int c = blockIdx.x;
  const int64_t n0 = (int64_t)c * L;
  const int64_t n1 = (n0 + L < N) ? (n0 + L) : N;
  const int nsub = n1 > n0 ? (int)((n1 - n0 + kWave - 1) / kWave) : 0;
  typedef float Vec4 __attribute__((ext_vector_type(4)));
  // where the lane's u-th 16-byte piece of a sub-tile lands in the padded tile
  int dst_off[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int e0 = (lane + u * kWave) * 4;
    const int r = e0 / E;
    dst_off[u] = r * ES + (e0 - r * E);
  }
  Vec4 pre[NU];
  auto request = [&](int ti) {
    const int64_t nb = n0 + (int64_t)ti * kWave;
    const int rows = (int)((n1 - nb < kWave) ? (n1 - nb) : kWave);
    const int nv = rows * E / 4;
    const Vec4* base = reinterpret_cast<const Vec4*>(y + (size_t)nb * E);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int i = lane + u * kWave;
      pre[u] = base[i < nv ? i : nv - 1];
    }
  };
  double acc[K], s0[K], s2[K];
#pragma unroll
  for (int k = 0; k < K; ++k) acc[k] = s0[k] = s2[k] = 0.0;
  if (wave < nsub) request(wave);
  for (int ti = wave; ti < nsub; ti += NWAVE) {
    const int64_t nb = n0 + (int64_t)ti * kWave;
    const int rows = (int)((n1 - nb < kWave) ? (n1 - nb) : kWave);
    const bool mine = lane < rows;
    // the row's spatial inputs: independent loads, in flight while the tile is staged
    double qv[K], ld[K], wv[K], sv = 1.0;
    int64_t gidx;
    {
      const int64_t n = nb + (mine ? lane : 0);
      const int64_t f = n / T;
      const int t = (int)(n - f * T);
      gidx = (f * K) * (int64_t)T + t;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        qv[k] = Q[gidx + (int64_t)k * T];
        ld[k] = lndet[f * K + k];
        wv[k] = weight ? weight[f * wb + k * wk + (int64_t)t * wt] : 1.0;
      }
      if (sal) sv = sal[n];
    }
    {
      const int nv = rows * E / 4;
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        if (lane + u * kWave < nv) {
          float* dst = tile + dst_off[u];
#pragma unroll
          for (int x = 0; x < 4; ++x) dst[x] = pre[u][x];
        }
      }
    }
    if (ti + NWAVE < nsub) request(ti + NWAVE);
    sweep_lds_order();
    // ---- E part: lane = row (the arithmetic of joint_sweep_kernel, same references)
    {
      double w[K];
#pragma unroll
      for (int k = 0; k < K; ++k) w[k] = 0.0;
      if (mine) {
        const float* row = tile + (size_t)lane * ES;
        double n2 = 0.0, a[K], d2[K];
#pragma unroll
        for (int k = 0; k < K; ++k) a[k] = d2[k] = 0.0;
        if (GAUSS) {
          for (int e = 0; e < E; ++e) {
            const double v = (double)row[e];
#pragma unroll
            for (int k = 0; k < K; ++k) {
              const double u = v - mean[k * E + e];  // gaussian.py:127-131
              a[k] = fma(u, u, a[k]);
            }
          }
#pragma unroll
          for (int k = 0; k < K; ++k) {
            d2[k] = a[k];
            a[k] *= prec[k] * prec[k];
          }
        } else {
          for (int e = 0; e < E; ++e) {
            const double v = (double)row[e];
            n2 = fma(v, v, n2);
#pragma unroll
            for (int k = 0; k < K; ++k) a[k] = fma(v, mean[k * E + e], a[k]);
          }
        }
        const double inv = GAUSS ? 0.0
                                 : ((n2 > 1e-280 && n2 < 1e280) ? fast_rsqrt(n2)
                                                                 : 1.0 / fmax(sqrt(n2), kTiny));
        double lp[K], mx = -1.79e308;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const double spec = GAUSS ? offset[k] - 0.5 * a[k]                // gaussian.py:132-136
                                    : fma(prec[k], a[k] * inv, offset[k]);  // von_mises_fisher.py:71-77
          const double lq = (qv[k] >= kTiny && qv[k] < 1.79e308) ? log_pos(qv[k]) : log(qv[k]);
          const double spat = -(double)D * lq - ld[k];                      // cacg.py:200-201
          lp[k] = spatial_weight * spat + spectral_weight * spec;
          mx = fmax(mx, lp[k]);
        }
        double g[K], den = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          g[k] = exp_nonpos(lp[k] - mx) * wv[k];  // mixture_model_utils.py:30-37
          den += g[k];
        }
        const double rden = fast_rcp(fmax(den, kTiny)) + (den - den);  // :43-47
#pragma unroll
        for (int k = 0; k < K; ++k) {
          double gm = g[k] * rden;
          if (eps != 0.0) gm = fmin(fmax(gm, eps), 1.0 - eps);  // :50-53
          G[gidx + (int64_t)k * T] = gm;
          const double wk2 = gm * sv;
          s0[k] += wk2;
          if (GAUSS) s2[k] = fma(wk2, d2[k], s2[k]);
          w[k] = wk2;
        }
      }
#pragma unroll
      for (int k = 0; k < KW; k += 2) {
        double2 p2;
        p2.x = w[k];
        p2.y = (k + 1 < K) ? w[k + 1 < K ? k + 1 : 0] : 0.0;
        *reinterpret_cast<double2*>(gam + lane * KW + k) = p2;
      }
    }
    sweep_lds_order();
    // ---- M part: lane = dimension; the row's weights are one broadcast read
    if (lane < E) {
#pragma unroll 4
      for (int r = 0; r < rows; ++r) {
        const double v = (double)tile[(size_t)r * ES + lane];
        double wk3[KW];
#pragma unroll
        for (int k = 0; k < KW; k += 2) {
          const double2 p2 = *reinterpret_cast<const double2*>(gam + r * KW + k);
          wk3[k] = p2.x;
          wk3[k + 1] = p2.y;
        }
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] = fma(wk3[k], v, acc[k]);
      }
    }
    sweep_lds_order();  // the next sub-tile overwrites what the M part has read
  }
  // ---- chunk partial: the four wavefronts in ascending order (bit-reproducible)
  if (lane < E) {
#pragma unroll
    for (int k = 0; k < K; ++k) red[((size_t)wave * K + k) * E + lane] = acc[k];
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const double t = wave_sum(s0[k]);
    if (lane == 0) red0[wave * K + k] = t;
    if (GAUSS) {
      const double t2 = wave_sum(s2[k]);
      if (lane == 0) red2[wave * K + k] = t2;
    }
  }
  __syncthreads();
  double* dst = part + (size_t)c * K * (E + 1);
  for (int i = tid; i < K * E; i += kFusedThreads) {
    const int k = i / E, dd = i - k * E;
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < NWAVE; ++w) t += red[((size_t)w * K + k) * E + dd];
    dst[k * (E + 1) + dd] = t;
    if (GAUSS) {
      double t2 = 0.0;  // 'spherical': the whole second moment travels in dimension 0
      if (dd == 0)
        for (int w = 0; w < NWAVE; ++w) t2 += red2[w * K + k];
      part2[(size_t)c * K * (E + 1) + k * (E + 1) + dd] = t2;
    }
  }
  if (tid < K) {
    double t = 0.0;
    for (int w = 0; w < NWAVE; ++w) t += red0[w * K + tid];
    dst[tid * (E + 1) + E] = t;
  }
}

// ---------------------------------------------------------------- M-step finalize
// One 1024-thread workgroup per mixture: ordered two-level sum over the chunk partials
// (slot = chunk index mod nslot, then over slots), then the model and -- so that the
// next E-step needs no separate launch -- the log-pdf offsets / precisions.
constexpr int kFinThreads = 1024;
constexpr int kFinLoads = 32;  // chunk partials in flight per thread (power of two): the partials were
                               // written by other XCDs, every dependent round is a ~2 us trip to the memory side

__device__ __forceinline__ double vmf_offset(int E, double conc, int lane) {
  // -log_norm = -(E/2 ln 2pi + ln ive(nu, k) + (|k| - nu ln k)) = -(E/2 ln 2pi + ln(I_nu(k)/k^nu))
  return -(0.5 * E * kLn2Pi + wave_log_bessel_over_power(0.5 * E - 1.0, conc, lane));
}

template <int KIND, int PASS>
__global__ void __launch_bounds__(kFinThreads)
    embed_finalize_kernel(const double* part, int C, int E, int K, double cmin, double cmax,
                          int weight_mode, double* den_buf, double* out_mean, double* out_scale,
                          double* out_weight, double* out_offset, double* out_prec) {
  extern __shared__ double sm[];  // tot [K][E+1], then red [nslot][W] (W < 1024) 
  const int64_t b = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = tid >> 6;
  const int W = K * (E + 1);
  double* tot = sm;
  double* red = sm + W;
  const double* pb = part + (size_t)b * C * W;
  if (W >= kFinThreads) {
    for (int i = tid; i < W; i += kFinThreads) {
      double t = 0.0;
      for (int c = 0; c < C; ++c) t += pb[(size_t)c * W + i];
      tot[i] = t;
    }
  } else {
    const int nslot = kFinThreads / W;
    const int slot = tid / W;
    const int i = tid - slot * W;
    if (slot < nslot) {
      // kFinLoads partials in flight per thread: the kernel is ONE workgroup walking C partials
      // through dependent L2 round trips (four in flight: 8 rounds = 10 us of a 39 us iteration)
      double t[kFinLoads];
#pragma unroll
      for (int u = 0; u < kFinLoads; ++u) t[u] = 0.0;
      for (int c = slot; c < C; c += kFinLoads * nslot) {
        double a[kFinLoads];
#pragma unroll
        for (int u = 0; u < kFinLoads; ++u) {
          const int cc = c + u * nslot;
          a[u] = pb[(size_t)(cc < C ? cc : slot) * W + i];  // clamped, masked below
        }
#pragma unroll
        for (int u = 0; u < kFinLoads; ++u) t[u] += (c + u * nslot < C) ? a[u] : 0.0;
      }
#pragma unroll
      for (int w = kFinLoads / 2; w >= 1; w /= 2)
#pragma unroll
        for (int u = 0; u < w; ++u) t[u] += t[u + w];
      red[slot * W + i] = t[0];
    }
    __syncthreads();
    for (int j = tid; j < W; j += kFinThreads) {
      double t = 0.0;
      for (int sl = 0; sl < nslot; ++sl) t += red[sl * W + j];
      tot[j] = t;
    }
  }
  __syncthreads();
  for (int k = wave; k < K; k += kFinThreads / kWave) {
    const double* row = tot + k * (E + 1);
    if (PASS == 0) {
      const double s0 = row[E];
      if (KIND == PBBSS_EMBED_VMF) {
        double n2 = 0.0;
        for (int d = lane; d < E; d += kWave) n2 = fma(row[d], row[d], n2);
        n2 = wave_sum(n2);
        const double norm = sqrt(n2);
        const double rn = 1.0 / fmax(norm, kTiny);  // Banerjee 2005 eq. 2.4
        for (int d = lane; d < E; d += kWave) out_mean[((size_t)b * K + k) * E + d] = row[d] * rn;
        const double rbar = norm / s0;                                         // eq. 2.5
        double conc = (rbar * E - rbar * rbar * rbar) / (1.0 - rbar * rbar);  // eq. 4.4
        conc = conc < cmin ? cmin : (conc > cmax ? cmax : conc);              // NaN stays NaN
        double off = 0.0;
        if (out_offset) off = vmf_offset(E, conc, lane);
        if (lane == 0) {
          out_scale[b * K + k] = conc;
          if (out_offset) {
            out_offset[b * K + k] = off;
            out_prec[b * K + k] = conc;
          }
        }
      } else {
        const double den = fmax(s0, kTiny);  // gaussian.py:160-163
        for (int d = lane; d < E; d += kWave) out_mean[((size_t)b * K + k) * E + d] = row[d] / den;
        if (lane == 0) den_buf[b * K + k] = den;
      }
    } else if (KIND == PBBSS_EMBED_GAUSS_DIAG) {
      // 'diagonal' (gaussian.py:176-179): one variance per dimension, out_scale (B, K, E)
      const double den = den_buf[b * K + k];
      for (int d = lane; d < E; d += kWave) out_scale[((size_t)b * K + k) * E + d] = row[d] / den;
    } else {
      double t = 0.0;
      for (int d = lane; d < E; d += kWave) t += row[d];
      t = wave_sum(t);
      if (lane == 0) {
        const double cv = t / (den_buf[b * K + k] * (double)E);  // 'spherical'
        out_scale[b * K + k] = cv;
        if (out_offset) {
          const double pc = 1.0 / sqrt(cv);
          out_offset[b * K + k] = -0.5 * E * kLn2Pi + (double)E * log(pc);
          out_prec[b * K + k] = pc;
        }
      }
    }
  }
  if (PASS == 0 && out_weight && weight_mode >= 0 && tid == 0) {
    if (weight_mode == 1) {
      for (int k = 0; k < K; ++k) out_weight[b * K + k] = 1.0 / K;
    } else {
      // estimate_mixture_weight with saliency: L1 unit norm over classes, eps 'where' 1e-10
      double t = 0.0;
      for (int k = 0; k < K; ++k) t += fabs(tot[k * (E + 1) + E]);
      if (t == 0.0) t = 1e-10;
      for (int k = 0; k < K; ++k) out_weight[b * K + k] = tot[k * (E + 1) + E] / t;
    }
  }
}

// Spherical Gaussian from ONE sweep (PASS 2 partials): mean = S1 / den, and with the shift c
//   sum_n w (y - mean)^2 = S2' - 2 (mean - c)(S1 - c S0) + (mean - c)^2 S0      per dimension.
__global__ void __launch_bounds__(kFinThreads)
    embed_finalize_single_kernel(const double* part, const double* part2, int C, int E, int K,
                                 const void* yr, int y_is_f64, int64_t N, int shift_first_row,
                                 double* out_mean, double* out_scale, double* out_offset,
                                 double* out_prec) {
  extern __shared__ double sm[];  // tot1 [K][E+1], tot2 [K][E+1]
  const int64_t b = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = tid >> 6;
  const int W = K * (E + 1);
  double* tot1 = sm;
  double* tot2 = sm + W;
  // Ordered two-level sum over the C chunk partials: slot = chunk index mod nslot, then over the
  // slots (as in embed_finalize_kernel).  One thread per value walking all C chunks was a chain
  // of C / 4 dependent L2 round trips: 22 us of the 98 us GCACGMM iteration.
  double* red = sm + 2 * W;  // [nslot][2 W]
  if (2 * W >= kFinThreads) {
    for (int i = tid; i < 2 * W; i += kFinThreads) {
      const double* p = (i < W ? part : part2) + (size_t)b * C * W + (i < W ? i : i - W);
      double t = 0.0;
      for (int c = 0; c < C; ++c) t += p[(size_t)c * W];
      sm[i] = t;
    }
  } else {
    const int nslot = kFinThreads / (2 * W);
    const int slot = tid / (2 * W);
    const int i = tid - slot * 2 * W;
    if (slot < nslot) {
      const double* p = (i < W ? part : part2) + (size_t)b * C * W + (i < W ? i : i - W);
      double t[kFinLoads];
#pragma unroll
      for (int u = 0; u < kFinLoads; ++u) t[u] = 0.0;
      for (int c = slot; c < C; c += kFinLoads * nslot) {
        double a[kFinLoads];
#pragma unroll
        for (int u = 0; u < kFinLoads; ++u) {
          const int cc = c + u * nslot;
          a[u] = p[(size_t)(cc < C ? cc : slot) * W];
        }
#pragma unroll
        for (int u = 0; u < kFinLoads; ++u) t[u] += (c + u * nslot < C) ? a[u] : 0.0;
      }
#pragma unroll
      for (int w = kFinLoads / 2; w >= 1; w /= 2)
#pragma unroll
        for (int u = 0; u < w; ++u) t[u] += t[u + w];
      red[slot * 2 * W + i] = t[0];
    }
    __syncthreads();
    for (int j = tid; j < 2 * W; j += kFinThreads) {
      double tt = 0.0;
      for (int sl = 0; sl < nslot; ++sl) tt += red[sl * 2 * W + j];
      sm[j] = tt;
    }
  }
  __syncthreads();
  for (int k = wave; k < K; k += kFinThreads / kWave) {
    const double* r1 = tot1 + k * (E + 1);
    const double* r2 = tot2 + k * (E + 1);
    const double s0 = r1[E];
    const double den = fmax(s0, kTiny);  // gaussian.py:160-163
    double acc = 0.0;
    for (int d = lane; d < E; d += kWave) {
      double cshift;
      if (shift_first_row) {
        cshift = y_is_f64 ? static_cast<const double*>(yr)[(size_t)b * N * E + d]
                          : (double)static_cast<const float*>(yr)[(size_t)b * N * E + d];
      } else {
        cshift = out_mean[((size_t)b * K + k) * E + d];  // previous mean (read before the write)
      }
      const double m = r1[d] / den;
      const double dm = m - cshift;
      acc += r2[d] - 2.0 * dm * (r1[d] - cshift * s0) + dm * dm * s0;
      out_mean[((size_t)b * K + k) * E + d] = m;
    }
    acc = wave_sum(acc);
    if (lane == 0) {
      const double cv = acc / (den * (double)E);  // 'spherical', gaussian.py:179-182
      out_scale[b * K + k] = cv;
      if (out_offset) {
        const double pc = 1.0 / sqrt(cv);
        out_offset[b * K + k] = -0.5 * E * kLn2Pi + (double)E * log(pc);
        out_prec[b * K + k] = pc;
      }
    }
  }
}

// ---------------------------------------------------------------- joint-model class weights
// mode 0 / 2: workgroup f: sum_t aff[f,k,t] sal[f,t]  -> tmp[f,k]; mode 0 normalises in place
__global__ void __launch_bounds__(kThreads)
    joint_rowsum_kernel(const double* aff, const double* sal, int K, int T, int normalize,
                        double* tmp, double* out) {
  __shared__ double red[kThreads / kWave][kEmbedMaxK];
  const int64_t f = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
  double s[kEmbedMaxK];
#pragma unroll
  for (int k = 0; k < kEmbedMaxK; ++k) s[k] = 0.0;
  for (int t = tid; t < T; t += kThreads) {
    const double sv = sal ? sal[(size_t)f * T + t] : 1.0;
#pragma unroll
    for (int k = 0; k < kEmbedMaxK; ++k)
      if (k < K) s[k] += aff[((size_t)f * K + k) * T + t] * sv;
  }
#pragma unroll
  for (int k = 0; k < kEmbedMaxK; ++k) {
    double t = wave_sum(s[k]);
    if (lane == 0) red[wave][k] = t;
  }
  __syncthreads();
  if (tid == 0) {
    double v[kEmbedMaxK], tot = 0.0;
    for (int k = 0; k < K; ++k) {
      v[k] = 0.0;
      for (int w = 0; w < kThreads / kWave; ++w) v[k] += red[w][k];
      tot += v[k];
    }
    for (int k = 0; k < K; ++k) {
      if (normalize) out[f * K + k] = v[k] / tot;  // gcacgmm.py:292-294
      else tmp[f * K + k] = v[k];
    }
  }
}

__global__ void joint_rows_to_class_kernel(const double* tmp, int64_t F, int K, int normalize,
                                           double* out) {
  // single workgroup: (-3,-1): sum over f of the row sums, normalised over k
  __shared__ double red[kThreads / kWave][kEmbedMaxK];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
  double s[kEmbedMaxK];
#pragma unroll
  for (int k = 0; k < kEmbedMaxK; ++k) s[k] = 0.0;
  for (int64_t f = tid; f < F; f += kThreads) {
#pragma unroll
    for (int k = 0; k < kEmbedMaxK; ++k)
      if (k < K) s[k] += tmp[f * K + k];
  }
#pragma unroll
  for (int k = 0; k < kEmbedMaxK; ++k) {
    double t = wave_sum(s[k]);
    if (lane == 0) red[wave][k] = t;
  }
  __syncthreads();
  if (tid == 0) {
    double v[kEmbedMaxK], tot = 0.0;
    for (int k = 0; k < K; ++k) {
      v[k] = 0.0;
      for (int w = 0; w < kThreads / kWave; ++w) v[k] += red[w][k];
      tot += v[k];
    }
    for (int k = 0; k < K; ++k) out[k] = normalize ? v[k] / tot : v[k];
  }
}

// out (K, T) raw class sums -> normalised over the classes, per column (after an all-reduce of
// the sums of a sharded fit)
__global__ void __launch_bounds__(kThreads) joint_normalize_kernel(double* out, int K, int T) {
  const int t = blockIdx.x * kThreads + threadIdx.x;
  if (t >= T) return;
  double tot = 0.0;
  for (int k = 0; k < K; ++k) tot += out[(size_t)k * T + t];
  for (int k = 0; k < K; ++k) out[(size_t)k * T + t] /= tot;
}

// mode 3 (-3,): sum over f, normalised over k  -> (K,T). One thread per frame walking all bins
// is 4 workgroups at T = 1000 and 445 us of load latency; the bins are cut into kColSlices
// slices (grid.y), the slice sums go to tmp (slice, K, T) and a second small kernel adds them
// in slice order and normalises.
__global__ void __launch_bounds__(kThreads)
    joint_colsum_part_kernel(const double* aff, const double* sal, int64_t F, int K, int T,
                             double* part) {
  const int t = blockIdx.x * kThreads + threadIdx.x;
  if (t >= T) return;
  const int64_t per = (F + gridDim.y - 1) / gridDim.y;
  const int64_t f0 = per * blockIdx.y;
  const int64_t f1 = f0 + per < F ? f0 + per : F;
  double s[kEmbedMaxK];
#pragma unroll
  for (int k = 0; k < kEmbedMaxK; ++k) s[k] = 0.0;
  for (int64_t f = f0; f < f1; ++f) {
    const double sv = sal ? sal[(size_t)f * T + t] : 1.0;
#pragma unroll
    for (int k = 0; k < kEmbedMaxK; ++k)
      if (k < K) s[k] += aff[((size_t)f * K + k) * T + t] * sv;
  }
#pragma unroll
  for (int k = 0; k < kEmbedMaxK; ++k)
    if (k < K) part[((size_t)blockIdx.y * K + k) * T + t] = s[k];
}

// 32 frames x 8 slice groups per workgroup: every thread adds its group's slices (c = g, g + 8,
// ...), the groups meet in LDS and are added in group order.
constexpr int kColFinFrames = 32;
constexpr int kColFinGroups = kThreads / kColFinFrames;
__global__ void __launch_bounds__(kThreads)
    joint_colsum_fin_kernel(const double* part, int slices, int K, int T, int normalize,
                            double* out) {
  __shared__ double red[kColFinGroups][kEmbedMaxK][kColFinFrames];
  const int tl = threadIdx.x % kColFinFrames, g = threadIdx.x / kColFinFrames;
  const int t = blockIdx.x * kColFinFrames + tl;
  double s[kEmbedMaxK];
#pragma unroll
  for (int k = 0; k < kEmbedMaxK; ++k) s[k] = 0.0;
  if (t < T) {
    for (int c = g; c < slices; c += kColFinGroups) {
#pragma unroll
      for (int k = 0; k < kEmbedMaxK; ++k)
        if (k < K) s[k] += part[((size_t)c * K + k) * T + t];
    }
  }
#pragma unroll
  for (int k = 0; k < kEmbedMaxK; ++k) red[g][k][tl] = s[k];
  __syncthreads();
  if (g != 0 || t >= T) return;
  double tot = 0.0;
#pragma unroll
  for (int k = 0; k < kEmbedMaxK; ++k) {
    if (k < K) {
      double v = 0.0;
      for (int gg = 0; gg < kColFinGroups; ++gg) v += red[gg][k][tl];
      s[k] = v;
      tot += v;
    }
  }
#pragma unroll
  for (int k = 0; k < kEmbedMaxK; ++k)
    if (k < K) out[(size_t)k * T + t] = normalize ? s[k] / tot : s[k];
}

__global__ void joint_fill_kernel(double* out, double v) { out[0] = v; }

inline int ok_or_hip() { return hipGetLastError() == hipSuccess ? PBBSS_OK : PBBSS_ERR_HIP; }

// sample slots of one fit workgroup: as many as 1024 threads hold, capped so that the
// slot-reduction buffer stays within 48 KiB of LDS
int fit_slots(int E, int K) {
  int S = kFitThreads / E;
  // two slot-reduction buffers (single-pass Gaussian), the slot sums and the weight staging of
  // kFitUnroll trips within the 64 KiB of dynamic LDS a launch gets without an attribute
  const int cap = 8000 / (K * (2 * E + 1 + kFitUnroll));
  if (S > cap) S = cap;
  return S < 1 ? 1 : S;
}

int fit_chunks(int64_t B, int64_t N, int E, int K) {
  const int S = fit_slots(E, K);
  int64_t want = 256 / (B < 256 ? B : 256);  // ~one 1024-thread workgroup per CU in total
  if (want < 1) want = 1;
  int64_t maxc = (N + (int64_t)S * 4 - 1) / ((int64_t)S * 4);  // >= 4 trips of one slot per chunk
  if (maxc < 1) maxc = 1;
  return (int)(want < maxc ? want : maxc);
}

}  // namespace

size_t embed_partial_doubles(int64_t B, int64_t N, int E, int K, int* chunks_out) {
  const int C = fit_chunks(B, N, E, K);
  if (chunks_out) *chunks_out = C;
  return 2 * (size_t)B * C * K * (E + 1) + (size_t)B * K;  // S1/S0, S2' (single pass), den
}

int launch_embed_prepare(const void* y, int y_is_f64, int64_t B, int64_t N, int E, int normalize,
                         void* yd, double* yr, hipStream_t s, double* rowscale) {
  if (E < 1 || E > kEmbedMaxE || B > 65535) return PBBSS_ERR_UNSUPPORTED;
  int R = 4096 / (E + 1);
  if (R > 64) R = 64;
  if (R < 1) R = 1;
  const size_t lds = ((size_t)R * (E + 1) + R) * sizeof(double);
  dim3 grid((unsigned)((N + R - 1) / R), (unsigned)B);
  if (normalize) {
    if (y_is_f64)
      hipLaunchKernelGGL((embed_prepare_kernel<double, true>), grid, dim3(kThreads), lds, s,
                         static_cast<const double*>(y), N, E, R, yd, yr, rowscale);
    else
      hipLaunchKernelGGL((embed_prepare_kernel<float, true>), grid, dim3(kThreads), lds, s,
                         static_cast<const float*>(y), N, E, R, yd, yr, rowscale);
  } else {
    if (y_is_f64)
      hipLaunchKernelGGL((embed_prepare_kernel<double, false>), grid, dim3(kThreads), lds, s,
                         static_cast<const double*>(y), N, E, R, yd, yr, rowscale);
    else
      hipLaunchKernelGGL((embed_prepare_kernel<float, false>), grid, dim3(kThreads), lds, s,
                         static_cast<const float*>(y), N, E, R, yd, yr, rowscale);
  }
  return ok_or_hip();
}

int launch_embed_offsets(int kind, int64_t BK, int E, const double* scale, double* offset,
                         double* prec, hipStream_t s) {
  const int per = kThreads / kWave;
  hipLaunchKernelGGL(embed_offsets_kernel, dim3((unsigned)((BK + per - 1) / per)), dim3(kThreads),
                     0, s, kind, BK, E, scale, offset, prec);
  return ok_or_hip();
}

namespace {
template <int KIND, int K, typename TS>
int estep_go(const void* yd, int64_t B, int64_t N, int E, const double* mean, const double* prec,
             const double* offset, const double* weight, double out_scale, int64_t Tin,
             double* out_lp, double* out_aff, hipStream_t s) {
  const size_t lds = ((size_t)K * E + 3 * K) * sizeof(double);
  dim3 grid((unsigned)((N + kThreads - 1) / kThreads), (unsigned)B);
  hipLaunchKernelGGL((embed_estep_kernel<KIND, K, TS>), grid, dim3(kThreads), lds, s,
                     static_cast<const TS*>(yd), N, E, mean, prec, offset, weight, out_scale, Tin,
                     out_lp, out_aff);
  return ok_or_hip();
}

template <int KIND, typename TS>
int estep_k(int K, const void* yd, int64_t B, int64_t N, int E, const double* mean,
            const double* prec, const double* offset, const double* weight, double out_scale,
            int64_t Tin, double* out_lp, double* out_aff, hipStream_t s) {
#define PBBSS_ESTEP_CASE(KK) \
  case KK: return estep_go<KIND, KK, TS>(yd, B, N, E, mean, prec, offset, weight, out_scale, Tin, out_lp, out_aff, s);
  switch (K) {
    PBBSS_ESTEP_CASE(1) PBBSS_ESTEP_CASE(2) PBBSS_ESTEP_CASE(3)
    PBBSS_ESTEP_CASE(4) PBBSS_ESTEP_CASE(5) PBBSS_ESTEP_CASE(6)
    PBBSS_ESTEP_CASE(7) PBBSS_ESTEP_CASE(8)
    default: return PBBSS_ERR_UNSUPPORTED;
  }
#undef PBBSS_ESTEP_CASE
}

// ---------------------------------------------------------------- diagonal Gaussian E-step
// DiagonalGaussian.log_pdf AS WRITTEN in the reference (gaussian.py:76-97): the (K, E) array
// of 1/sqrt(variance) is handed to einsum('...dD,...nD->...nd') as ONE K x E matrix shared by
// all classes, i.e.
//   white[k, n, j] = sum_e pc[j, e] (y[n, e] - mean[k, e])          j = 0..K-1
//   log_pdf[k, n]  = -E/2 ln 2pi + sum_e ln pc[k, e] - 1/2 sum_j white[k, n, j]^2
// (not the textbook per-class whitening; the drop-in reproduces what the reference computes).
// With u_j(n) = sum_e pc[j, e] y[n, e] and c[j, k] = sum_e pc[j, e] mean[k, e]:
//   white[k, n, j] = u_j(n) - c[j, k].
// consts layout: pc (K, E) | off (K) | c (K, K)
__global__ void __launch_bounds__(kThreads) diag_consts_kernel(const double* mean,
                                                               const double* cov, int K, int E,
                                                               double* consts) {
  double* pc = consts;
  double* off = consts + (size_t)K * E;
  double* cm = off + K;
  for (int i = threadIdx.x; i < K * E; i += kThreads) pc[i] = 1.0 / sqrt(cov[i]);
  // one wavefront per output value, lanes over the E dimensions (a thread per value walked E
  // dependent loads of values other threads had just written: 15 us for 12 numbers)
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  for (int i = wave; i < K + K * K; i += kThreads / kWave) {
    double t = 0.0;
    if (i < K) {
      for (int e = lane; e < E; e += kWave) t += -0.5 * log(cov[i * E + e]);  // ln(1 / sqrt(cov))
      t = wave_sum(t);
      if (lane == 0) off[i] = -0.5 * E * kLn2Pi + t;
    } else {
      const int j = (i - K) / K, k = (i - K) - j * K;
      for (int e = lane; e < E; e += kWave) t = fma(1.0 / sqrt(cov[j * E + e]), mean[k * E + e], t);
      t = wave_sum(t);
      if (lane == 0) cm[j * K + k] = t;
    }
  }
}

template <typename TS>
__global__ void __launch_bounds__(kThreads) diag_estep_kernel(const TS* yd, int64_t N, int E,
                                                              int K, const double* consts,
                                                              double out_scale, int64_t Tin,
                                                              double* out_lp) {
  extern __shared__ double sm[];  // pc (K, E) | off (K) | c (K, K)
  const int nconst = K * E + K + K * K;
  for (int i = threadIdx.x; i < nconst; i += kThreads) sm[i] = consts[i];
  __syncthreads();
  const double* pc = sm;
  const double* off = sm + K * E;
  const double* cm = off + K;
  const int64_t n = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (n >= N) return;
  double u[kEmbedMaxK];
  for (int j = 0; j < K; ++j) u[j] = 0.0;
  for (int e = 0; e < E; ++e) {
    const double yv = (double)yd[(size_t)e * N + n];
    for (int j = 0; j < K; ++j) u[j] = fma(pc[j * E + e], yv, u[j]);
  }
  for (int k = 0; k < K; ++k) {
    double q = 0.0;
    for (int j = 0; j < K; ++j) {
      const double w = u[j] - cm[j * K + k];
      q = fma(w, w, q);
    }
    out_lp[aff_index(0, k, n, K, N, Tin)] = out_scale * (off[k] - 0.5 * q);
  }
}

// ---------------------------------------------------------------- layout changes for the
// full-covariance kernels (gauss_full.hip works on (K, N) arrays, the joint models on (F, K, T))
// w[k, f*T + t] = aff[f, k, t] * (sal ? sal[f, t] : 1)      'fkt->k,ft' (gcacgmm.py:298-302)
__global__ void __launch_bounds__(kThreads) fkt_to_kn_kernel(const double* aff, const double* sal,
                                                             int64_t F, int K, int T,
                                                             double* out) {
  const int64_t total = F * K * T;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kThreads) {
    const int64_t t = i % T, k = (i / T) % K, f = i / ((int64_t)T * K);
    out[(size_t)k * F * T + f * T + t] = aff[i] * (sal ? sal[f * T + t] : 1.0);
  }
}
// out[f, k, t] = scale * lp[k, f*T + t]
__global__ void __launch_bounds__(kThreads) kn_to_fkt_kernel(const double* lp, double scale,
                                                             int64_t F, int K, int T,
                                                             double* out) {
  const int64_t total = F * K * T;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * kThreads) {
    const int64_t t = i % T, k = (i / T) % K, f = i / ((int64_t)T * K);
    out[i] = scale * lp[(size_t)k * F * T + f * T + t];
  }
}

template <int K, typename TS>
int fit_go(int kind, const void* yr, int64_t B, int64_t N, int E, const double* aff, int64_t Tin,
           const double* sal, double cmin, double cmax, int weight_mode, double* part,
           double* out_mean, double* out_scale, double* out_weight, double* out_offset,
           double* out_prec, int single_pass, const double* rowscale, hipStream_t s,
           const PartialReduce* reduce) {
  int C = 0;
  const size_t np = embed_partial_doubles(B, N, E, K, &C);
  const size_t nsum = (size_t)B * C * K * (E + 1);  // one set of chunk partials
  auto all_ranks = [&](double* buf, size_t count) -> int {
    return reduce ? reduce->fn(reduce->ctx, buf, count, s) : PBBSS_OK;
  };
  // the first-row shift of the one-sweep variance differs from rank to rank: sharded fits take
  // the previous mean (identical on all ranks) or, before there is one, two sweeps
  if (reduce && single_pass == 2) single_pass = 0;
  double* den_buf = part + np - (size_t)B * K;
  const int S = fit_slots(E, K);
  int64_t L = (N + C - 1) / C;
  L = (L + S - 1) / S * S;
  const size_t Wv = (size_t)K * (E + 1);
  const size_t lds_fit = (2 * (size_t)S * K * E + (size_t)S * K + (size_t)kFitUnroll * S * K) * sizeof(double);  // + weight staging
  const size_t lds_fin = (Wv + (Wv < (size_t)kFinThreads ? (kFinThreads / Wv) * Wv : 0)) * sizeof(double);
  dim3 grid((unsigned)C, (unsigned)B);
  if (kind == PBBSS_EMBED_GAUSS_SPHERICAL && single_pass != 0) {
    // single sweep: shift = previous mean (single_pass == 1) or the first row (== 2)
    double* part2 = part + (size_t)B * C * K * (E + 1);
    const int first = single_pass == 2;
    hipLaunchKernelGGL((embed_fit_kernel<K, TS, 2>), grid, dim3(kFitThreads), lds_fit, s,
                       static_cast<const TS*>(yr), N, E, S, C, L, aff, Tin, sal, out_mean, part,
                       part2, first, (const double*)nullptr);
    if (int rc = all_ranks(part, 2 * nsum); rc != PBBSS_OK) return rc;  // part2 follows part
    hipLaunchKernelGGL(embed_finalize_single_kernel, dim3((unsigned)B), dim3(kFinThreads),
                       (2 * Wv + (2 * Wv < (size_t)kFinThreads ? (size_t)kFinThreads : 0)) * sizeof(double),
                       s, part, part2, C, E, K, yr,
                       (int)std::is_same<TS, double>::value, N, first, out_mean, out_scale,
                       out_offset, out_prec);
    return ok_or_hip();
  }
  hipLaunchKernelGGL((embed_fit_kernel<K, TS, 0>), grid, dim3(kFitThreads), lds_fit, s,
                     static_cast<const TS*>(yr), N, E, S, C, L, aff, Tin, sal,
                     (const double*)nullptr, part, (double*)nullptr, 0, rowscale);
  if (int rc = all_ranks(part, nsum); rc != PBBSS_OK) return rc;
  if (kind == PBBSS_EMBED_VMF) {
    hipLaunchKernelGGL((embed_finalize_kernel<PBBSS_EMBED_VMF, 0>), dim3((unsigned)B),
                       dim3(kFinThreads), lds_fin, s, part, C, E, K, cmin, cmax, weight_mode,
                       den_buf, out_mean, out_scale, out_weight, out_offset, out_prec);
    return ok_or_hip();
  }
  hipLaunchKernelGGL((embed_finalize_kernel<PBBSS_EMBED_GAUSS_SPHERICAL, 0>), dim3((unsigned)B),
                     dim3(kFinThreads), lds_fin, s, part, C, E, K, cmin, cmax, weight_mode, den_buf,
                     out_mean, out_scale, out_weight, (double*)nullptr, (double*)nullptr);
  hipLaunchKernelGGL((embed_fit_kernel<K, TS, 1>), grid, dim3(kFitThreads), lds_fit, s,
                     static_cast<const TS*>(yr), N, E, S, C, L, aff, Tin, sal, out_mean, part,
                     (double*)nullptr, 0, (const double*)nullptr);
  if (int rc = all_ranks(part, nsum); rc != PBBSS_OK) return rc;
  if (kind == PBBSS_EMBED_GAUSS_DIAG) {
    hipLaunchKernelGGL((embed_finalize_kernel<PBBSS_EMBED_GAUSS_DIAG, 1>), dim3((unsigned)B),
                       dim3(kFinThreads), lds_fin, s, part, C, E, K, cmin, cmax, -1, den_buf,
                       out_mean, out_scale, (double*)nullptr, (double*)nullptr, (double*)nullptr);
    return ok_or_hip();
  }
  hipLaunchKernelGGL((embed_finalize_kernel<PBBSS_EMBED_GAUSS_SPHERICAL, 1>), dim3((unsigned)B),
                     dim3(kFinThreads), lds_fin, s, part, C, E, K, cmin, cmax, -1, den_buf, out_mean,
                     out_scale, (double*)nullptr, out_offset, out_prec);
  return ok_or_hip();
}

template <typename TS>
int fit_k(int K, int kind, const void* yr, int64_t B, int64_t N, int E, const double* aff,
          int64_t Tin, const double* sal, double cmin, double cmax, int weight_mode, double* part,
          double* out_mean, double* out_scale, double* out_weight, double* out_offset,
          double* out_prec, int single_pass, const double* rowscale, hipStream_t s,
          const PartialReduce* reduce) {
#define PBBSS_FIT_CASE(KK) \
  case KK: return fit_go<KK, TS>(kind, yr, B, N, E, aff, Tin, sal, cmin, cmax, weight_mode, part, out_mean, out_scale, out_weight, out_offset, out_prec, single_pass, rowscale, s, reduce);
  switch (K) {
    PBBSS_FIT_CASE(1) PBBSS_FIT_CASE(2) PBBSS_FIT_CASE(3)
    PBBSS_FIT_CASE(4) PBBSS_FIT_CASE(5) PBBSS_FIT_CASE(6)
    PBBSS_FIT_CASE(7) PBBSS_FIT_CASE(8)
    default: return PBBSS_ERR_UNSUPPORTED;
  }
#undef PBBSS_FIT_CASE
}
}  // namespace

int launch_embed_estep(int kind, const void* yd, int y_is_f64, int64_t B, int64_t N, int E, int K,
                       const double* mean, const double* prec, const double* offset,
                       const double* weight, double out_scale, int64_t Tin, double* out_lp,
                       double* out_aff, hipStream_t s) {
  if (E < 1 || E > kEmbedMaxE || B > 65535) return PBBSS_ERR_UNSUPPORTED;
  if (kind == PBBSS_EMBED_VMF) {
    return y_is_f64 ? estep_k<PBBSS_EMBED_VMF, double>(K, yd, B, N, E, mean, prec, offset, weight,
                                                       out_scale, Tin, out_lp, out_aff, s)
                    : estep_k<PBBSS_EMBED_VMF, float>(K, yd, B, N, E, mean, prec, offset, weight,
                                                      out_scale, Tin, out_lp, out_aff, s);
  }
  if (kind == PBBSS_EMBED_GAUSS_SPHERICAL) {
    return y_is_f64 ? estep_k<PBBSS_EMBED_GAUSS_SPHERICAL, double>(
                          K, yd, B, N, E, mean, prec, offset, weight, out_scale, Tin, out_lp,
                          out_aff, s)
                    : estep_k<PBBSS_EMBED_GAUSS_SPHERICAL, float>(
                          K, yd, B, N, E, mean, prec, offset, weight, out_scale, Tin, out_lp,
                          out_aff, s);
  }
  return PBBSS_ERR_UNSUPPORTED;
}

int launch_embed_fit(int kind, const void* yr, int y_is_f64, int64_t B, int64_t N, int E, int K,
                     const double* aff, int64_t Tin, const double* sal, double cmin, double cmax,
                     int weight_mode, double* part, double* out_mean, double* out_scale,
                     double* out_weight, double* out_offset, double* out_prec, int single_pass,
                     hipStream_t s, const double* rowscale, const PartialReduce* reduce) {
  if (E < 1 || E > kEmbedMaxE || B > 65535) return PBBSS_ERR_UNSUPPORTED;
  if (kind != PBBSS_EMBED_VMF && kind != PBBSS_EMBED_GAUSS_SPHERICAL && kind != PBBSS_EMBED_GAUSS_DIAG)
    return PBBSS_ERR_UNSUPPORTED;
  if (rowscale && kind != PBBSS_EMBED_VMF) return PBBSS_ERR_INVALID_ARG;
  if (kind == PBBSS_EMBED_GAUSS_DIAG) single_pass = 0;  // two sweeps, as the reference
  return y_is_f64 ? fit_k<double>(K, kind, yr, B, N, E, aff, Tin, sal, cmin, cmax, weight_mode,
                                  part, out_mean, out_scale, out_weight, out_offset, out_prec,
                                  single_pass, rowscale, s, reduce)
                  : fit_k<float>(K, kind, yr, B, N, E, aff, Tin, sal, cmin, cmax, weight_mode,
                                 part, out_mean, out_scale, out_weight, out_offset, out_prec,
                                 single_pass, rowscale, s, reduce);
}

// ---- fused vMF sweep (vmf_em_kernel) + finalize: one EM iteration of the vMF mixture ----------
namespace {
struct FusedPlan {
  int R, C;
  int64_t L;
  size_t lds;
  bool ok;
};
FusedPlan fused_plan(int64_t B, int64_t N, int E, int K, int y_is_f64) {
  FusedPlan p{};
  const size_t esz = y_is_f64 ? 8 : 4;
  const int ES = E | 1;
  const int S = kFusedThreads / E;
  for (int R : {256, 128, 64}) {
    const int KW = (K + 1) & ~1;
    const size_t lds = ((size_t)R * KW + (size_t)S * K * E + (size_t)(kFusedThreads / kWave) * K) * 8 +
                       (size_t)R * ES * esz;
    if (lds <= 64 * 1024) {
      p.R = R;
      p.lds = lds;
      p.ok = true;
      break;
    }
  }
  if (!p.ok || E > kFusedThreads) {
    p.ok = false;
    return p;
  }
  // ~two workgroups per CU in total (256 / 384 / 768 / 1024 measured slower: 35.9 / 35.9 / 35.5 /
  // 40.5 against 32.0 us per iteration at N = 256 500, E = 40)
  int64_t want = 512 / (B < 512 ? B : 512);
  if (want < 1) want = 1;
  int64_t maxc = (N + p.R - 1) / p.R;
  p.C = (int)(want < maxc ? want : maxc);
  p.L = (N + p.C - 1) / p.C;
  p.L = (p.L + p.R - 1) / p.R * p.R;
  return p;
}
}  // namespace

size_t vmf_fused_partial_doubles(int64_t B, int64_t N, int E, int K, int y_is_f64) {
  const FusedPlan p = fused_plan(B, N, E, K, y_is_f64);
  if (!p.ok) return 0;
  return (size_t)B * p.C * K * (E + 1) + (size_t)B * K;  // chunk partials, den
}

int launch_vmf_em(const void* y, int y_is_f64, int64_t B, int64_t N, int E, int K,
                  const double* gamma, const double* sal, double cmin, double cmax,
                  int weight_mode, double* part, double* mean, double* conc, double* weight,
                  double* offset, double* prec, double* out_aff, int accumulate, hipStream_t s) {
  const FusedPlan p = fused_plan(B, N, E, K, y_is_f64);
  if (!p.ok || B > 65535 || K < 1 || K > kEmbedMaxK) return PBBSS_ERR_UNSUPPORTED;
  dim3 grid((unsigned)p.C, (unsigned)B);
  // 16-byte vector loads need E * sizeof(element) to be a multiple of 16 (rows then start on a
  // vector boundary, the tile base is R-row aligned) and a 16-byte aligned array
  const int vw = y_is_f64 ? 2 : 4;
  const bool vec = (E % vw == 0) && (reinterpret_cast<uintptr_t>(y) % 16 == 0);
#define PBBSS_VMF_GO(KK, TT, VV)                                                                  \
  hipLaunchKernelGGL((vmf_em_kernel<KK, TT, VV>), grid, dim3(kFusedThreads), p.lds, s,            \
                     static_cast<const TT*>(y), N, E, p.R, p.C, p.L, gamma, mean, prec, offset,   \
                     weight, sal, part, out_aff)
#define PBBSS_VMF_EM(KK)                                                                          \
  case KK:                                                                                        \
    if (y_is_f64) {                                                                               \
      if (vec) PBBSS_VMF_GO(KK, double, true); else PBBSS_VMF_GO(KK, double, false);              \
    } else {                                                                                      \
      if (vec) PBBSS_VMF_GO(KK, float, true); else PBBSS_VMF_GO(KK, float, false);                \
    }                                                                                             \
    break;
  switch (K) {
    PBBSS_VMF_EM(1) PBBSS_VMF_EM(2) PBBSS_VMF_EM(3) PBBSS_VMF_EM(4) PBBSS_VMF_EM(5) PBBSS_VMF_EM(6)
    PBBSS_VMF_EM(7) PBBSS_VMF_EM(8)
    default: return PBBSS_ERR_UNSUPPORTED;
  }
#undef PBBSS_VMF_EM
#undef PBBSS_VMF_GO
  if (hipGetLastError() != hipSuccess) return PBBSS_ERR_HIP;
  if (!accumulate) return PBBSS_OK;  // predict only: the partials are not used
  const size_t Wv = (size_t)K * (E + 1);
  const size_t lds_fin = (Wv + (Wv < (size_t)kFinThreads ? (kFinThreads / Wv) * Wv : 0)) * sizeof(double);
  // (A first reduction level inside the sweep -- the last workgroup of every group of 32 chunks
  // sums the group's partials, the finalize then reads 16 instead of 512 -- was built and
  // measured: the sweep grew by the 4 us the finalize lost, 32.5 vs 32.0 us per iteration.)
  double* den_buf = part + (size_t)B * p.C * K * (E + 1);
  hipLaunchKernelGGL((embed_finalize_kernel<PBBSS_EMBED_VMF, 0>), dim3((unsigned)B),
                     dim3(kFinThreads), lds_fin, s, part, p.C, E, K, cmin, cmax, weight_mode,
                     den_buf, mean, conc, weight, offset, prec);
  return ok_or_hip();
}

// ---- persistent vMF mixture for many small mixtures (vmf_bin_em_kernel) ------------------------
size_t vmf_bin_lds_bytes(int64_t N, int E, int K, int y_is_f64) {
  if (E < 1 || E > kFusedThreads || N < 1 || N > 65536) return 0;
  const int KW = (K + 1) & ~1, ES = E | 1, S = kFusedThreads / E;
  const size_t dbl = (size_t)N * KW + (size_t)S * K * E + (size_t)(kFusedThreads / kWave) * K +
                     (size_t)K * E + (size_t)K * (E + 1) + 3 * (size_t)K;
  return dbl * 8 + (size_t)N * ES * (y_is_f64 ? 8 : 4) + 16;
}

int launch_vmf_bin_em(const void* y, int y_is_f64, int64_t B, int64_t N, int E, int K,
                      int iterations, const double* gamma, const double* sal,
                      const double* in_mean, const double* in_conc, const double* in_weight,
                      double cmin, double cmax, int weight_mode, double* mean, double* conc,
                      double* weight, double* out_aff, size_t lds_limit, hipStream_t s) {
  const size_t lds = vmf_bin_lds_bytes(N, E, K, y_is_f64);
  if (lds == 0 || lds > lds_limit || K < 1 || K > kEmbedMaxK || B > 2147483647LL)
    return PBBSS_ERR_UNSUPPORTED;
#define PBBSS_VB_GO(KK, TT)                                                                        \
  {                                                                                                \
    auto kfn = vmf_bin_em_kernel<KK, TT>;                                                          \
    if (lds > 64 * 1024 &&                                                                         \
        hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),                                    \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)   \
      return PBBSS_ERR_HIP;                                                                        \
    hipLaunchKernelGGL(kfn, dim3((unsigned)B), dim3(kFusedThreads), lds, s,                        \
                       static_cast<const TT*>(y), N, E, iterations, gamma, sal, in_mean, in_conc,  \
                       in_weight, cmin, cmax, weight_mode, mean, conc, weight, out_aff);           \
  }
#define PBBSS_VB_K(KK)                                                                             \
  case KK:                                                                                         \
    if (y_is_f64) PBBSS_VB_GO(KK, double) else PBBSS_VB_GO(KK, float)                              \
    break;
  switch (K) {
    PBBSS_VB_K(1) PBBSS_VB_K(2) PBBSS_VB_K(3) PBBSS_VB_K(4) PBBSS_VB_K(5) PBBSS_VB_K(6)
    PBBSS_VB_K(7) PBBSS_VB_K(8)
    default: return PBBSS_ERR_UNSUPPORTED;
  }
#undef PBBSS_VB_K
#undef PBBSS_VB_GO
  return ok_or_hip();
}

// ---- rotated joint loop: sweep (posteriors + spectral sums) and its finalize -------------------
namespace {
FusedPlan joint_sweep_plan(int64_t N, int E, int K, int y_is_f64, bool gauss) {
  FusedPlan p{};
  const size_t esz = y_is_f64 ? 8 : 4;
  const int ES = E | 1;
  if (E > kFusedThreads || E < 1) return p;
  const int S = kFusedThreads / E;
  const int KW = (K + 1) & ~1;
  static const int rows_env = [] {  // development knob: rows per tile of the sweep
    const char* v = getenv("PBBSS_JOINT_SWEEP_ROWS");
    return v ? atoi(v) : 0;
  }();
  for (int R : {256, 128, 64}) {
    if (rows_env > 0 && R > rows_env) continue;
    size_t tile_bytes = (size_t)R * ES * esz;
    const size_t red_bytes = (gauss ? 2 : 1) * (size_t)S * K * E * 8;  // aliases the tile
    if (tile_bytes < red_bytes) tile_bytes = red_bytes;
    const size_t lds = ((size_t)R * KW + (size_t)(kFusedThreads / kWave) * K) * 8 + tile_bytes;
    if (lds <= 64 * 1024) {
      p.R = R;
      p.lds = lds;
      p.ok = true;
      break;
    }
  }
  if (!p.ok) return p;
  static const int want_env = [] {  // development knob: chunks (= workgroups) of the sweep
    const char* v = getenv("PBBSS_JOINT_SWEEP_CHUNKS");
    return v ? atoi(v) : 0;
  }();
  int64_t want = want_env > 0 ? want_env : 512;
  int64_t maxc = (N + p.R - 1) / p.R;
  p.C = (int)(want < maxc ? want : maxc);
  p.L = (N + p.C - 1) / p.C;
  p.L = (p.L + p.R - 1) / p.R * p.R;
  return p;
}
}  // namespace

bool joint_sweep_supported(int kind, int64_t N, int E, int K, int y_is_f64) {
  if (kind != PBBSS_EMBED_VMF && kind != PBBSS_EMBED_GAUSS_SPHERICAL) return false;
  if (K < 1 || K > kEmbedMaxK) return false;
  return joint_sweep_plan(N, E, K, y_is_f64, kind != PBBSS_EMBED_VMF).ok;
}

void joint_sweep_chunks(int kind, int64_t N, int E, int K, int y_is_f64, int* chunks) {
  const FusedPlan p = joint_sweep_plan(N, E, K, y_is_f64, kind != PBBSS_EMBED_VMF);
  *chunks = p.ok ? p.C : 0;
}

// chunk partials S1/S0 (+ S2' for the Gaussian) and the finalize's denominators
size_t joint_sweep_partial_doubles(int kind, int64_t N, int E, int K, int y_is_f64) {
  const FusedPlan p = joint_sweep_plan(N, E, K, y_is_f64, kind != PBBSS_EMBED_VMF);
  if (!p.ok) return 0;
  return 2 * (size_t)p.C * K * (E + 1) + (size_t)K;
}

int launch_joint_sweep(int kind, const void* y, int y_is_f64, int64_t F, int T, int E, int K, int D,
                       const double* Q, const double* lndet, const double* weight, int64_t wb,
                       int64_t wk, int64_t wt, const double* mean, const double* prec,
                       const double* offset, double spatial_weight, double spectral_weight,
                       const double* sal, double eps, double* G, double* part, hipStream_t s) {
  const int64_t N = F * (int64_t)T;
  const bool gauss = kind != PBBSS_EMBED_VMF;
  const FusedPlan p = joint_sweep_plan(N, E, K, y_is_f64, gauss);
  if (!p.ok || !joint_sweep_supported(kind, N, E, K, y_is_f64)) return PBBSS_ERR_UNSUPPORTED;
  double* part2 = part + (size_t)p.C * K * (E + 1);
  dim3 grid((unsigned)p.C);
  {
    // wave-private sweep (joint_sweep_wave_kernel): float32 rows of whole 16-byte pieces
    static const bool wave_off = [] {
      const char* v = getenv("PBBSS_JOINT_SWEEP_WAVE");
      return v && atoi(v) == 0;
    }();
    const int KW = (K + 1) & ~1;
    const size_t wlds = (size_t)(kFusedThreads / kWave) * ((size_t)kWave * (E | 1) * 4 + (size_t)kWave * KW * 8) +
                        ((size_t)(kFusedThreads / kWave) * K * E + 2 * (size_t)(kFusedThreads / kWave) * K) * 8;
    if (!wave_off && !y_is_f64 && E % 4 == 0 && E >= 4 && E <= 64 && K >= 2 && K <= 4 &&
        reinterpret_cast<uintptr_t>(y) % 16 == 0 && p.L % kWave == 0 && wlds <= 64 * 1024) {
#define PBBSS_JW_GO(KIND, KK, NU)                                                                  \
  hipLaunchKernelGGL((joint_sweep_wave_kernel<KIND, KK, NU>), grid, dim3(kFusedThreads), wlds, s,  \
                     static_cast<const float*>(y), N, E, p.L, T, D, Q, lndet, weight, wb, wk, wt,  \
                     mean, prec, offset, spatial_weight, spectral_weight, sal, eps, G, part, part2)
#define PBBSS_JW_N(KIND, KK)                                                                       \
  if (E <= 40) PBBSS_JW_GO(KIND, KK, 10); else PBBSS_JW_GO(KIND, KK, 16)
#define PBBSS_JW_K(KK)                                                                             \
  case KK:                                                                                         \
    if (gauss) { PBBSS_JW_N(PBBSS_EMBED_GAUSS_SPHERICAL, KK); } else { PBBSS_JW_N(PBBSS_EMBED_VMF, KK); } \
    break;
      switch (K) { PBBSS_JW_K(2) PBBSS_JW_K(3) PBBSS_JW_K(4) }
#undef PBBSS_JW_K
#undef PBBSS_JW_N
#undef PBBSS_JW_GO
      return ok_or_hip();
    }
  }
  const int vw = y_is_f64 ? 2 : 4;
  const bool vec = (E % vw == 0) && (reinterpret_cast<uintptr_t>(y) % 16 == 0);
#define PBBSS_JS_GO(KIND, KK, TT, VV)                                                              \
  hipLaunchKernelGGL((joint_sweep_kernel<KIND, KK, TT, VV>), grid, dim3(kFusedThreads), p.lds, s,  \
                     static_cast<const TT*>(y), N, E, p.R, p.C, p.L, T, D, Q, lndet, weight, wb,   \
                     wk, wt, mean, prec, offset, spatial_weight, spectral_weight, sal, eps, G,     \
                     part, part2)
#define PBBSS_JS_T(KIND, KK)                                                                       \
  if (y_is_f64) {                                                                                  \
    if (vec) PBBSS_JS_GO(KIND, KK, double, true); else PBBSS_JS_GO(KIND, KK, double, false);       \
  } else {                                                                                         \
    if (vec) PBBSS_JS_GO(KIND, KK, float, true); else PBBSS_JS_GO(KIND, KK, float, false);         \
  }
#define PBBSS_JS_K(KK)                                                                             \
  case KK:                                                                                         \
    if (gauss) { PBBSS_JS_T(PBBSS_EMBED_GAUSS_SPHERICAL, KK) } else { PBBSS_JS_T(PBBSS_EMBED_VMF, KK) } \
    break;
  switch (K) {
    PBBSS_JS_K(1) PBBSS_JS_K(2) PBBSS_JS_K(3) PBBSS_JS_K(4) PBBSS_JS_K(5) PBBSS_JS_K(6)
    PBBSS_JS_K(7) PBBSS_JS_K(8)
    default: return PBBSS_ERR_UNSUPPORTED;
  }
#undef PBBSS_JS_K
#undef PBBSS_JS_T
#undef PBBSS_JS_GO
  return ok_or_hip();
}

// the model of the spectral half from the sweep's partials (+ the constants the next sweep
// reads); `reduce`: sum the partials over the ranks first (bins sharded, pbbss_mix_opts.sharded)
int launch_joint_sweep_finalize(int kind, const void* y, int y_is_f64, int64_t N, int E, int K,
                                double cmin, double cmax, double* part, double* mean,
                                double* scale, double* offset, double* prec, hipStream_t s,
                                const PartialReduce* reduce) {
  const bool gauss = kind != PBBSS_EMBED_VMF;
  const FusedPlan p = joint_sweep_plan(N, E, K, y_is_f64, gauss);
  if (!p.ok) return PBBSS_ERR_UNSUPPORTED;
  const size_t nsum = (size_t)p.C * K * (E + 1);
  if (reduce) {
    if (int rc = reduce->fn(reduce->ctx, part, gauss ? 2 * nsum : nsum, s); rc != PBBSS_OK) return rc;
  }
  const size_t Wv = (size_t)K * (E + 1);
  if (gauss) {
    hipLaunchKernelGGL(embed_finalize_single_kernel, dim3(1), dim3(kFinThreads),
                       (2 * Wv + (2 * Wv < (size_t)kFinThreads ? (size_t)kFinThreads : 0)) * sizeof(double),
                       s, part, part + nsum, p.C, E, K, y, y_is_f64, N, 0, mean, scale, offset, prec);
    return ok_or_hip();
  }
  const size_t lds_fin = (Wv + (Wv < (size_t)kFinThreads ? (kFinThreads / Wv) * Wv : 0)) * sizeof(double);
  double* den_buf = part + 2 * nsum;
  hipLaunchKernelGGL((embed_finalize_kernel<PBBSS_EMBED_VMF, 0>), dim3(1), dim3(kFinThreads),
                     lds_fin, s, part, p.C, E, K, cmin, cmax, -1, den_buf, mean, scale,
                     (double*)nullptr, offset, prec);
  return ok_or_hip();
}

static int joint_colsum_slices(int64_t F) { return F < 64 ? (int)(F < 1 ? 1 : F) : 64; }

size_t joint_weight_tmp_doubles(int mode, int64_t F, int K, int T) {
  const size_t rows = (size_t)F * K;
  const size_t cols = mode == 3 ? (size_t)joint_colsum_slices(F) * K * T : 0;
  return rows > cols ? rows : cols;
}

int launch_joint_weight(int mode, const double* aff, const double* sal, int64_t F, int K, int T,
                        double* tmp, double* out_weight, hipStream_t s,
                        const PartialReduce* reduce) {
  if (K < 1 || K > kEmbedMaxK) return PBBSS_ERR_UNSUPPORTED;
  switch (mode) {
    case 0:
      hipLaunchKernelGGL(joint_rowsum_kernel, dim3((unsigned)F), dim3(kThreads), 0, s, aff, sal, K,
                         T, 1, tmp, out_weight);
      break;
    case 1:
      hipLaunchKernelGGL(joint_fill_kernel, dim3(1), dim3(1), 0, s, out_weight, 1.0 / K);
      break;
    case 2:
      hipLaunchKernelGGL(joint_rowsum_kernel, dim3((unsigned)F), dim3(kThreads), 0, s, aff, sal, K,
                         T, 0, tmp, out_weight);
      hipLaunchKernelGGL(joint_rows_to_class_kernel, dim3(1), dim3(kThreads), 0, s, tmp, F, K,
                         reduce ? 0 : 1, out_weight);
      if (reduce) {
        if (int rc = reduce->fn(reduce->ctx, out_weight, (size_t)K, s); rc != PBBSS_OK) return rc;
        hipLaunchKernelGGL(joint_normalize_kernel, dim3(1), dim3(kThreads), 0, s, out_weight, K, 1);
      }
      break;
    case 3:
    {
      const int slices = joint_colsum_slices(F);
      const unsigned gx = (unsigned)((T + kThreads - 1) / kThreads);
      hipLaunchKernelGGL(joint_colsum_part_kernel, dim3(gx, (unsigned)slices), dim3(kThreads), 0, s,
                         aff, sal, F, K, T, tmp);
      hipLaunchKernelGGL(joint_colsum_fin_kernel,
                         dim3((unsigned)((T + kColFinFrames - 1) / kColFinFrames)), dim3(kThreads), 0,
                         s, tmp, slices, K, T, reduce ? 0 : 1, out_weight);
      if (reduce) {
        if (int rc = reduce->fn(reduce->ctx, out_weight, (size_t)K * T, s); rc != PBBSS_OK)
          return rc;
        hipLaunchKernelGGL(joint_normalize_kernel, dim3((unsigned)((T + kThreads - 1) / kThreads)),
                           dim3(kThreads), 0, s, out_weight, K, T);
      }
      break;
    }
    case 4:
      hipLaunchKernelGGL(joint_fill_kernel, dim3(1), dim3(1), 0, s, out_weight, 1.0);
      break;
    default: return PBBSS_ERR_INVALID_ARG;
  }
  return ok_or_hip();
}

namespace {
__global__ void first_row_f64_kernel(const void* y, int y_is_f64, int E, double* out) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d < E)
    out[d] = y_is_f64 ? static_cast<const double*>(y)[d] : (double)static_cast<const float*>(y)[d];
}

__global__ void or_status_kernel(const int32_t* src, int32_t* dst) { dst[0] |= src[0]; }
}  // namespace
int launch_first_row_f64(const void* y, int y_is_f64, int E, double* out, hipStream_t s) {
  hipLaunchKernelGGL(first_row_f64_kernel, dim3((unsigned)((E + 63) / 64)), dim3(64), 0, s, y,
                     y_is_f64, E, out);
  return ok_or_hip();
}

int launch_or_status(const int32_t* src, int32_t* dst, hipStream_t s) {
  hipLaunchKernelGGL(or_status_kernel, dim3(1), dim3(1), 0, s, src, dst);
  return ok_or_hip();
}

size_t diag_consts_doubles(int K, int E) { return (size_t)K * E + K + (size_t)K * K; }

int launch_diag_estep(const void* yd, int y_is_f64, int64_t N, int E, int K, const double* mean,
                      const double* cov, double out_scale, int64_t Tin, double* consts,
                      double* out_lp, hipStream_t s) {
  if (E < 1 || E > kEmbedMaxE || K < 1 || K > kEmbedMaxK) return PBBSS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(diag_consts_kernel, dim3(1), dim3(kThreads), 0, s, mean, cov, K, E, consts);
  const size_t lds = diag_consts_doubles(K, E) * sizeof(double);
  const unsigned grid = (unsigned)((N + kThreads - 1) / kThreads);
  if (y_is_f64) {
    hipLaunchKernelGGL(diag_estep_kernel<double>, dim3(grid), dim3(kThreads), lds, s,
                       static_cast<const double*>(yd), N, E, K, consts, out_scale, Tin, out_lp);
  } else {
    hipLaunchKernelGGL(diag_estep_kernel<float>, dim3(grid), dim3(kThreads), lds, s,
                       static_cast<const float*>(yd), N, E, K, consts, out_scale, Tin, out_lp);
  }
  return ok_or_hip();
}

int launch_fkt_to_kn(const double* aff, const double* sal, int64_t F, int K, int T, double* out,
                     hipStream_t s) {
  const int64_t total = F * K * T;
  const unsigned grid = (unsigned)((total + kThreads - 1) / kThreads < 4096 ? (total + kThreads - 1) / kThreads : 4096);
  hipLaunchKernelGGL(fkt_to_kn_kernel, dim3(grid ? grid : 1), dim3(kThreads), 0, s, aff, sal, F, K,
                     T, out);
  return ok_or_hip();
}

int launch_kn_to_fkt(const double* lp, double scale, int64_t F, int K, int T, double* out,
                     hipStream_t s) {
  const int64_t total = F * K * T;
  const unsigned grid = (unsigned)((total + kThreads - 1) / kThreads < 4096 ? (total + kThreads - 1) / kThreads : 4096);
  hipLaunchKernelGGL(kn_to_fkt_kernel, dim3(grid ? grid : 1), dim3(kThreads), 0, s, lp, scale, F, K,
                     T, out);
  return ok_or_hip();
}

}  // namespace pbbss
