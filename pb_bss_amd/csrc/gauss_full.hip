// Full-covariance Gaussians on real embeddings for gfx950.
//
// Reference: distribution/gaussian.py:17-56 (Gaussian: sklearn precision Cholesky, log_pdf),
// :152-193 (GaussianTrainer._fit, covariance_type='full').
//
// The weighted scatter  sum_n w_kn (y_n - mean_k)(y_n - mean_k)^T  is the one GEMM-shaped
// piece of the whole library (E ~ 40: a 40 x N by N x 40 product per class), so it runs on the
// FP64 matrix pipe: with the augmented, shifted vector z_n = [y_n - c ; 1 ; 0 ...] (c = first
// row of the mixture, P = 16 NT entries) ONE symmetric Gram matrix
//     G_k = sum_n w_kn z_n z_n^T = [[S2', S1'], [S1'^T, S0]]
// carries the second moment about c, the first moment about c and the weight sum; the finalize
// turns them into mean and covariance about the mean (c within the data's spread: no
// cancellation to speak of).  Tiles are v_mfma_f64_16x16x4_f64: lane l feeds A[i = l % 16]
// [k = l / 16] and B[k = l / 16][j = l % 16] -- here sample k of a group of four, entries i / j
// of two 16-blocks of z -- and holds D[4 r + l / 16][l % 16] in accumulator register r
// (layout verified on the device by tools/ubench/mfma64.hip).  FP64 MFMA shares the FP64
// vector datapath on gfx950 (no second pipe) but a tile costs one issue slot
// for 2 048 flops, and the cross-lane reduction comes for free.
//
// Factorisation and log-pdf as the reference writes them:  cov = L L^T (Cholesky, lower),
// X = L^-1,  "precision Cholesky" P = X^T,  white_n = P (y_n - mean)  (gaussian.py:46-50: the
// einsum applies P, not P^T),  log_pdf = -E/2 ln 2pi + sum_d ln P_dd - 1/2 |white_n|^2, i.e.
// the quadratic form of  Mq = P^T P = X X^T.
#include <hip/hip_runtime.h>
#include "gauss_full.hpp"
#include "pbbss_dev.hpp"

namespace pbbss {
namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int kGfThreads = 256;
constexpr int kGfWaves = kGfThreads / kWave;
constexpr double kLn2PiGf = 1.8378770664093453;

inline int gf_ok() { return hipGetLastError() == hipSuccess ? PBBSS_OK : PBBSS_ERR_HIP; }

// ------------------------------------------------------------------ scatter (MFMA)
// grid (C, K, B); every wave of a workgroup owns a contiguous run of samples and writes its own
// partial: part[((b K + k) C 4 + c 4 + wave)][tile][r * 64 + lane], tiles (ti <= tj) row-major.
template <int NT, typename TS>
__global__ void __launch_bounds__(kGfThreads)
    gf_scatter_kernel(const TS* __restrict__ y, int64_t N, int E, int K,
                      const double* __restrict__ aff, const double* __restrict__ sal, int C,
                      int64_t Lw, double* __restrict__ part) {
  constexpr int NTT = NT * (NT + 1) / 2;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = blockIdx.x, k = blockIdx.y;
  const int64_t b = blockIdx.z;
  const int i = lane & 15, smp = lane >> 4;
  const TS* yb = y + (size_t)b * N * E;
  const double* wk = aff + ((size_t)b * K + k) * N;
  double shift[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int dim = 16 * t + i;
    shift[t] = (dim < E) ? (double)yb[dim] : 0.0;  // c = row 0 of the mixture
  }
  double4_t acc[NTT];
#pragma unroll
  for (int x = 0; x < NTT; ++x) acc[x] = double4_t{0.0, 0.0, 0.0, 0.0};
  const int64_t n0 = ((int64_t)c * kGfWaves + wave) * Lw;
  const int64_t n1 = (n0 + Lw < N) ? n0 + Lw : N;
  for (int64_t n = n0 + smp; n < n1 + smp; n += 4) {  // the four lane groups stay together
    const bool ok = n < n1;
    const int64_t nc = ok ? n : n0;
    double w = ok ? wk[nc] : 0.0;
    if (sal) w *= sal[(size_t)b * N + nc];  // affiliation * saliency (gmm.py:160)
    double v[NT], a[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int dim = 16 * t + i;
      double z = 0.0;
      if (dim < E) z = (double)yb[(size_t)nc * E + dim] - shift[t];
      if (dim == E) z = 1.0;
      v[t] = ok ? z : 0.0;
      a[t] = w * v[t];
    }
    int x = 0;
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
      for (int tj = ti; tj < NT; ++tj) {
        acc[x] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ti], v[tj], acc[x], 0, 0, 0);
        ++x;
      }
  }
  double* dst = part + ((((size_t)b * K + k) * C + c) * kGfWaves + wave) * (size_t)NTT * 256;
#pragma unroll
  for (int x = 0; x < NTT; ++x)
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[(size_t)x * 256 + r * 64 + lane] = acc[x][r];
}

// ------------------------------------------------------------------ small dense helpers (LDS)
// In-place lower Cholesky of the E x E matrix a (row stride ld).  Returns 0 or 1 + the index of
// the first non-positive pivot (LAPACK dpotrf INFO), uniform over the workgroup.
__device__ int gf_cholesky(double* a, int E, int ld, int tid, int* info_sm) {
  if (tid == 0) *info_sm = 0;
  __syncthreads();
  for (int j = 0; j < E; ++j) {
    if (tid == 0) {
      const double d = a[j * ld + j];
      if (!(d > 0.0) || !(d < 1.79e308)) {
        if (*info_sm == 0) *info_sm = j + 1;
        a[j * ld + j] = 1.0;
      } else {
        a[j * ld + j] = sqrt(d);
      }
    }
    __syncthreads();
    const double piv = a[j * ld + j];
    for (int r = j + 1 + tid; r < E; r += kGfThreads) a[r * ld + j] /= piv;
    __syncthreads();
    const int m = E - j - 1;  // trailing update of the lower triangle
    for (int idx = tid; idx < m * m; idx += kGfThreads) {
      const int r = j + 1 + idx / m, cc = j + 1 + idx % m;
      if (cc <= r) a[r * ld + cc] -= a[r * ld + j] * a[cc * ld + j];
    }
    __syncthreads();
  }
  return *info_sm;
}

// x = L^-1 (lower), one thread per column, forward substitution
__device__ void gf_tri_inverse(const double* l, double* x, int E, int ld, int tid) {
  for (int cidx = tid; cidx < E; cidx += kGfThreads) {
    for (int r = 0; r < E; ++r) {
      double s = (r == cidx) ? 1.0 : 0.0;
      for (int m = cidx; m < r; ++m) s -= l[r * ld + m] * x[m * ld + cidx];
      x[r * ld + cidx] = (r < cidx) ? 0.0 : s / l[r * ld + r];
    }
  }
  __syncthreads();
}

// cov (E x E, global) -> Mq = X X^T (global, E x E) and offset = -E/2 ln 2pi - sum ln L_dd;
// lds: 2 E (E + 1) doubles + 1 int.  Returns the Cholesky info.
__device__ int gf_factor(const double* cov, int E, double* lds, int* info_sm, double* out_mq,
                         double* out_offset, int tid) {
  const int ld = E + 1;
  double* l = lds;
  double* x = lds + (size_t)E * ld;
  for (int idx = tid; idx < E * E; idx += kGfThreads) {
    const int r = idx / E, cc = idx % E;
    // force the symmetry the reference's LAPACK call assumes (it reads the lower triangle)
    l[r * ld + cc] = (cc <= r) ? cov[(size_t)r * E + cc] : 0.0;
  }
  __syncthreads();
  const int info = gf_cholesky(l, E, ld, tid, info_sm);
  gf_tri_inverse(l, x, E, ld, tid);
  for (int idx = tid; idx < E * E; idx += kGfThreads) {
    const int r = idx / E, cc = idx % E;
    double s = 0.0;
    const int m1 = (r < cc) ? r : cc;  // X is lower triangular: X_rm = 0 for m > r
    for (int m = 0; m <= m1; ++m) s += x[r * ld + m] * x[cc * ld + m];
    out_mq[(size_t)r * E + cc] = s;
  }
  if (tid == 0) {
    double sl = 0.0;
    for (int d = 0; d < E; ++d) sl += log(l[d * ld + d]);
    *out_offset = -0.5 * E * kLn2PiGf - sl;  // sum_d ln P_dd = -sum_d ln L_dd
  }
  __syncthreads();
  return info;
}

// ------------------------------------------------------------------ finalize
// One workgroup per (b, k): ordered sum of the NP wave partials (slot-parallel like the other
// finalize kernels), G -> mean, covariance (gaussian.py:155-190), optionally the factorisation.
template <int NT, typename TS>
__global__ void __launch_bounds__(kGfThreads)
    gf_finalize_kernel(const double* __restrict__ part, int NP, const TS* __restrict__ y, int64_t N,
                       int E, int K, double* __restrict__ out_mean, double* __restrict__ out_cov,
                       double* out_mq, double* out_offset, double* out_s0,
                       int32_t* out_status) {
  constexpr int NTT = NT * (NT + 1) / 2;
  constexpr int P = 16 * NT;
  extern __shared__ double sm[];
  double* G = sm;  // [P][P + 1]
  __shared__ int info_sm;
  const int tid = threadIdx.x, lane = tid & 63;
  const int k = blockIdx.x;
  const int64_t b = blockIdx.y;
  const double* pb = part + (((size_t)b * K + k) * NP) * (size_t)NTT * 256;
  // element e of a tile: e = r * 64 + l  ->  row 4 r + l / 16, column l % 16
  for (int idx = tid; idx < NTT * 256; idx += kGfThreads) {
    double t[4] = {0.0, 0.0, 0.0, 0.0};
    int p = 0;
    for (; p + 3 < NP; p += 4) {
      double a0 = pb[(size_t)p * NTT * 256 + idx], a1 = pb[(size_t)(p + 1) * NTT * 256 + idx];
      double a2 = pb[(size_t)(p + 2) * NTT * 256 + idx], a3 = pb[(size_t)(p + 3) * NTT * 256 + idx];
      t[0] += a0;
      t[1] += a1;
      t[2] += a2;
      t[3] += a3;
    }
    for (; p < NP; ++p) t[0] += pb[(size_t)p * NTT * 256 + idx];
    const double tot = (t[0] + t[1]) + (t[2] + t[3]);
    const int x = idx >> 8, e = idx & 255;
    int ti = 0, tj = 0, cnt = 0;  // x-th tile of the upper triangle, row-major
    for (int a = 0; a < NT; ++a)
      for (int bb = a; bb < NT; ++bb) {
        if (cnt == x) {
          ti = a;
          tj = bb;
        }
        ++cnt;
      }
    const int r = e >> 6, l = e & 63;
    const int row = 16 * ti + 4 * r + (l >> 4), col = 16 * tj + (l & 15);
    G[row * (P + 1) + col] = tot;
    if (ti != tj) G[col * (P + 1) + row] = tot;
  }
  __syncthreads();
  const TS* yb = y + (size_t)b * N * E;
  const double s0 = G[E * (P + 1) + E];
  const double den = fmax(s0, kTiny);  // gaussian.py:160-163
  if (out_s0 && tid == 0) out_s0[(size_t)b * K + k] = s0;
  double* mean = out_mean + ((size_t)b * K + k) * E;
  double* cov = out_cov + ((size_t)b * K + k) * (size_t)E * E;
  for (int d = tid; d < E; d += kGfThreads) mean[d] = (double)yb[d] + G[E * (P + 1) + d] / den;
  for (int idx = tid; idx < E * E; idx += kGfThreads) {
    const int r = idx / E, cc = idx % E;
    // sum w (y - mean)(y - mean)^T with y - mean = (y - c) - m',  m' = S1' / den
    const double s1r = G[E * (P + 1) + r], s1c = G[E * (P + 1) + cc];
    const double mr = s1r / den, mc = s1c / den;
    // computed once per pair and mirrored: exactly symmetric (FMA contraction would break the
    // commutativity of an expression evaluated on both sides)
    if (r <= cc) {
      const double v = (G[r * (P + 1) + cc] - (mr * s1c + s1r * mc) + (mr * mc) * s0) / den;
      cov[(size_t)r * E + cc] = v;
      cov[(size_t)cc * E + r] = v;
    }
  }
  __syncthreads();
  if (out_mq) {
    __threadfence_block();
    int info = gf_factor(cov, E, sm, &info_sm, out_mq + ((size_t)b * K + k) * (size_t)E * E,
                         out_offset + (size_t)b * K + k, tid);
    if (info && tid == 0 && out_status) atomicOr(out_status, (int32_t)PBBSS_ST_NOT_POSDEF);
  }
  (void)lane;
}

// factorisation of given covariances: one workgroup per (b, k)
__global__ void __launch_bounds__(kGfThreads)
    gf_factor_kernel(const double* __restrict__ cov, int E, double* out_mq, double* out_offset,
                     int32_t* out_status) {
  extern __shared__ double sm[];
  __shared__ int info_sm;
  const int64_t bk = blockIdx.x;
  int info = gf_factor(cov + (size_t)bk * E * E, E, sm, &info_sm, out_mq + (size_t)bk * E * E,
                       out_offset + bk, threadIdx.x);
  if (info && threadIdx.x == 0 && out_status) atomicOr(out_status, (int32_t)PBBSS_ST_NOT_POSDEF);
}

// ------------------------------------------------------------------ log-pdf / E-step (MFMA)
// One wavefront per 16 samples and class: Z = Mq_k D^T by 16x16x4 tiles (A = Mq block, B = the
// samples' centred vectors), q_n = sum_i d_ni Z_in; D staged per workgroup in LDS.
// grid (ceil(N / 64), B): a workgroup = 4 waves = 64 samples; classes looped.
template <int NT, typename TS>
__global__ void __launch_bounds__(kGfThreads)
    gf_logpdf_kernel(const TS* __restrict__ y, int64_t N, int E, int K,
                     const double* __restrict__ mean, const double* __restrict__ mq,
                     const double* __restrict__ offset, const double* __restrict__ weight,
                     double* __restrict__ out_lp, double* __restrict__ out_aff) {
  constexpr int P = 16 * NT;
  extern __shared__ double sm[];
  double* M = sm;                    // [P][P + 1]  Mq_k, zero padded
  double* Dm = sm + P * (P + 1);     // [64][P + 1] centred samples, zero padded
  double* lp = Dm + 64 * (P + 1);    // [K][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t b = blockIdx.y;
  const int64_t nb = (int64_t)blockIdx.x * 64;
  const TS* yb = y + (size_t)b * N * E;
  for (int k = 0; k < K; ++k) {
    const double* mu = mean + ((size_t)b * K + k) * E;
    const double* mk = mq + ((size_t)b * K + k) * (size_t)E * E;
    __syncthreads();
    for (int idx = tid; idx < P * P; idx += kGfThreads) {
      const int r = idx / P, cc = idx % P;
      M[r * (P + 1) + cc] = (r < E && cc < E) ? mk[(size_t)r * E + cc] : 0.0;
    }
    for (int idx = tid; idx < 64 * P; idx += kGfThreads) {
      const int s = idx / P, d = idx % P;
      const int64_t n = nb + s;
      Dm[s * (P + 1) + d] = (n < N && d < E) ? (double)yb[(size_t)n * E + d] - mu[d] : 0.0;
    }
    __syncthreads();
    // wave w: samples 16 w .. 16 w + 15; lane: i = l % 16, g = l / 16
    const int i = lane & 15, g = lane >> 4;
    const double* ds = Dm + (16 * wave) * (P + 1);
    double qpart = 0.0;
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
      double4_t z = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < P; kk += 4) {
        const double a = M[(16 * ti + i) * (P + 1) + kk + g];  // A[i][k = g]
        const double bv = ds[i * (P + 1) + kk + g];            // B[k = g][j = i]: sample i, dim kk+g
        z = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, z, 0, 0, 0);
      }
      // z[r] = Z[out dim 16 ti + 4 r + g][sample i]
#pragma unroll
      for (int r = 0; r < 4; ++r) qpart = fma(ds[i * (P + 1) + 16 * ti + 4 * r + g], z[r], qpart);
    }
    qpart += __shfl_xor(qpart, 16, 64);
    qpart += __shfl_xor(qpart, 32, 64);
    if (g == 0) lp[k * 64 + 16 * wave + i] = offset[(size_t)b * K + k] - 0.5 * qpart;
  }
  __syncthreads();
  if (tid < 64) {
    const int64_t n = nb + tid;
    if (n < N) {
      double mx = -1.79e308;
      for (int k = 0; k < K; ++k) mx = fmax(mx, lp[k * 64 + tid]);
      if (out_lp)
        for (int k = 0; k < K; ++k) out_lp[((size_t)b * K + k) * N + n] = lp[k * 64 + tid];
      if (out_aff) {  // mixture_model_utils.py:30-47, affiliation_eps = 0
        double den = 0.0;
        for (int k = 0; k < K; ++k) den += exp(lp[k * 64 + tid] - mx) * weight[(size_t)b * K + k];
        den = fmax(den, kTiny);
        for (int k = 0; k < K; ++k)
          out_aff[((size_t)b * K + k) * N + n] =
              exp(lp[k * 64 + tid] - mx) * weight[(size_t)b * K + k] / den;
      }
    }
  }
}

// estimate_mixture_weight with saliency (mixture_model_utils.py:192-201): L1 unit norm of the
// masked affiliation sums over the classes, eps 'where' 1e-10; mode 1: uniform 1 / K
__global__ void gf_weights_kernel(const double* s0, int64_t B, int K, int mode, double* out) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  if (mode == 1) {
    for (int k = 0; k < K; ++k) out[b * K + k] = 1.0 / K;
    return;
  }
  double t = 0.0;
  for (int k = 0; k < K; ++k) t += fabs(s0[b * K + k]);
  if (t == 0.0) t = 1e-10;
  for (int k = 0; k < K; ++k) out[b * K + k] = s0[b * K + k] / t;
}

int gf_chunks(int64_t B, int K, int64_t N) {
  // ~1024 waves in flight in total, at least 64 samples per wave
  int64_t c = 256 / (B * K < 256 ? B * K : 256);
  const int64_t maxc = (N + 4 * 64 - 1) / (4 * 64);
  if (c > maxc) c = maxc;
  return (int)(c < 1 ? 1 : c);
}

template <int NT, typename TS>
int gf_fit_go(const void* y, int64_t B, int64_t N, int E, int K, const double* w,
              const double* sal, double* part, double* out_mean, double* out_cov, double* out_mq,
              double* out_offset, double* out_s0, int32_t* out_status, hipStream_t s) {
  constexpr int P = 16 * NT;
  const int C = gf_chunks(B, K, N);
  int64_t Lw = (N + (int64_t)C * kGfWaves - 1) / ((int64_t)C * kGfWaves);
  Lw = (Lw + 3) / 4 * 4;
  hipLaunchKernelGGL((gf_scatter_kernel<NT, TS>), dim3((unsigned)C, (unsigned)K, (unsigned)B),
                     dim3(kGfThreads), 0, s, static_cast<const TS*>(y), N, E, K, w, sal, C, Lw,
                     part);
  size_t ldsd = (size_t)P * (P + 1);
  if (out_mq && 2 * (size_t)E * (E + 1) > ldsd) ldsd = 2 * (size_t)E * (E + 1);
  const size_t lds = ldsd * sizeof(double);
  auto kfn = gf_finalize_kernel<NT, TS>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return PBBSS_ERR_HIP;
  hipLaunchKernelGGL(kfn, dim3((unsigned)K, (unsigned)B), dim3(kGfThreads), lds, s, part,
                     C * kGfWaves, static_cast<const TS*>(y), N, E, K, out_mean, out_cov, out_mq,
                     out_offset, out_s0, out_status);
  return gf_ok();
}

template <int NT, typename TS>
int gf_logpdf_go(const void* y, int64_t B, int64_t N, int E, int K, const double* mean,
                 const double* mq, const double* offset, const double* weight, double* out_lp,
                 double* out_aff, hipStream_t s) {
  constexpr int P = 16 * NT;
  const size_t lds = ((size_t)P * (P + 1) + 64 * (size_t)(P + 1) + (size_t)K * 64) * sizeof(double);
  auto kfn = gf_logpdf_kernel<NT, TS>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return PBBSS_ERR_HIP;
  hipLaunchKernelGGL(kfn, dim3((unsigned)((N + 63) / 64), (unsigned)B), dim3(kGfThreads), lds, s,
                     static_cast<const TS*>(y), N, E, K, mean, mq, offset, weight, out_lp, out_aff);
  return gf_ok();
}

}  // namespace

size_t gauss_full_partial_doubles(int64_t B, int64_t N, int E, int K) {
  const int NT = (E + 1 + 15) / 16;
  const int C = gf_chunks(B, K, N);
  return (size_t)B * K * C * kGfWaves * (size_t)(NT * (NT + 1) / 2) * 256;
}

#define PBBSS_GF_DISPATCH(FN, ...)                                                      \
  switch ((E + 1 + 15) / 16) {                                                          \
    case 1: return y_is_f64 ? FN<1, double>(__VA_ARGS__) : FN<1, float>(__VA_ARGS__);   \
    case 2: return y_is_f64 ? FN<2, double>(__VA_ARGS__) : FN<2, float>(__VA_ARGS__);   \
    case 3: return y_is_f64 ? FN<3, double>(__VA_ARGS__) : FN<3, float>(__VA_ARGS__);   \
    case 4: return y_is_f64 ? FN<4, double>(__VA_ARGS__) : FN<4, float>(__VA_ARGS__);   \
    default: return PBBSS_ERR_UNSUPPORTED;                                              \
  }

int launch_gauss_full_fit(const void* y, int y_is_f64, int64_t B, int64_t N, int E, int K,
                          const double* weights, const double* sal, double* part,
                          double* out_mean, double* out_cov, double* out_mq, double* out_offset,
                          double* out_s0, int32_t* out_status, hipStream_t s) {
  if (E < 1 || E > kGaussFullMaxE || K < 1 || B < 1 || B > 65535 || K > 65535)
    return PBBSS_ERR_UNSUPPORTED;
  PBBSS_GF_DISPATCH(gf_fit_go, y, B, N, E, K, weights, sal, part, out_mean, out_cov, out_mq,
                    out_offset, out_s0, out_status, s)
}

int launch_gauss_full_weights(const double* s0, int64_t B, int K, int mode, double* out_weight,
                              hipStream_t s) {
  hipLaunchKernelGGL(gf_weights_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, s, s0, B, K,
                     mode, out_weight);
  return gf_ok();
}

int launch_gauss_full_factor(const double* cov, int64_t BK, int E, double* out_mq,
                             double* out_offset, int32_t* out_status, hipStream_t s) {
  if (E < 1 || E > kGaussFullMaxE) return PBBSS_ERR_UNSUPPORTED;
  const size_t lds = 2 * (size_t)E * (E + 1) * sizeof(double);
  hipLaunchKernelGGL(gf_factor_kernel, dim3((unsigned)BK), dim3(kGfThreads), lds, s, cov, E, out_mq,
                     out_offset, out_status);
  return gf_ok();
}

int launch_gauss_full_logpdf(const void* y, int y_is_f64, int64_t B, int64_t N, int E, int K,
                             const double* mean, const double* mq, const double* offset,
                             const double* weight, double* out_lp, double* out_aff,
                             hipStream_t s) {
  if (E < 1 || E > kGaussFullMaxE || K < 1 || K > 64 || B > 65535) return PBBSS_ERR_UNSUPPORTED;
  if (out_aff && !weight) return PBBSS_ERR_INVALID_ARG;
  PBBSS_GF_DISPATCH(gf_logpdf_go, y, B, N, E, K, mean, mq, offset, weight, out_lp, out_aff, s)
}
#undef PBBSS_GF_DISPATCH

}  // namespace pbbss
