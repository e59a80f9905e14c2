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
// the quadratic form of  P^T P = X X^T.  The kernels keep P itself (upper triangular).
#include <hip/hip_runtime.h>
#include <cstdlib>
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
// grid (C, K, B); every wave of a workgroup owns a contiguous run of samples; the workgroup
// writes one partial: part[(b K + k) C + c][tile][r * 64 + lane], tiles (ti <= tj) row-major.
template <int NT, typename TS>
__global__ void __launch_bounds__(kGfThreads)
    gf_scatter_kernel(const TS* __restrict__ y, int64_t N, int E, int K,
                      const double* __restrict__ aff, const double* __restrict__ sal, int C,
                      int64_t Lw, double* __restrict__ part, const double* __restrict__ gshift) {
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
    // c = row 0 of the mixture, or the caller's shift (bins sharded over ranks: every rank must
    // subtract the SAME c for the partial sums to add up)
    shift[t] = (dim < E) ? (gshift ? gshift[dim] : (double)yb[dim]) : 0.0;
  }
  double4_t acc[NTT];
#pragma unroll
  for (int x = 0; x < NTT; ++x) acc[x] = double4_t{0.0, 0.0, 0.0, 0.0};
  const int64_t n0 = ((int64_t)c * kGfWaves + wave) * Lw;
  const int64_t n1 = (n0 + Lw < N) ? n0 + Lw : N;
  // U groups of four samples per trip.  Software-pipelined: the loads of trip t + 1 are issued
  // before the tiles of trip t (one wave per SIMD: nothing else hides the HBM round trip; a
  // load-then-compute loop ran at 6 TFLOP/s of tiles, with all loads of a trip up front 13).
  constexpr int U = 4;
  // Loads are UNCONDITIONAL at clamped addresses and land in raw registers; masks and
  // arithmetic are applied when the values are consumed.  (A guarded load whose value is
  // converted inside the guard compiles to branch + load + s_waitcnt vmcnt(0): one full memory
  // round trip per element.)
  const double* salb = sal ? sal + (size_t)b * N : wk;  // dummy pointer keeps the load uniform
  const double salmask = sal ? 1.0 : 0.0;
  int dimc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) dimc[t] = (16 * t + i < E) ? 16 * t + i : 0;
  double wn[U], sn[U];
  TS rn[U][NT];
  auto fetch = [&](int64_t nb) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t n = nb + 4 * u + smp;
      const int64_t nc = (n < n1) ? n : n0;
      wn[u] = wk[nc];
      sn[u] = salb[nc];
#pragma unroll
      for (int t = 0; t < NT; ++t) rn[u][t] = yb[(size_t)nc * E + dimc[t]];
    }
  };
  if (n0 < n1) fetch(n0);
  for (int64_t nb = n0; nb < n1; nb += 4 * U) {  // the four lane groups stay together
    double w[U], v[U][NT];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = nb + 4 * u + smp < n1;
      // affiliation * saliency (gmm.py:160); without saliency the factor is exactly 1
      const double sv = salmask * sn[u] + (1.0 - salmask);
      w[u] = ok ? wn[u] * sv : 0.0;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int dim = 16 * t + i;
        double z = (double)rn[u][t] - shift[t];
        z = (dim < E) ? z : ((dim == E) ? 1.0 : 0.0);
        v[u][t] = ok ? z : 0.0;
      }
    }
    if (nb + 4 * U < n1) fetch(nb + 4 * U);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int x = 0;
#pragma unroll
      for (int ti = 0; ti < NT; ++ti) {
        const double a = w[u] * v[u][ti];
#pragma unroll
        for (int tj = ti; tj < NT; ++tj) {
          acc[x] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, v[u][tj], acc[x], 0, 0, 0);
          ++x;
        }
      }
    }
  }
  // the four wave partials of the workgroup are combined through LDS (fixed order) before they
  // leave the CU: a quarter of the partial traffic for the ordered reduction that follows
  extern __shared__ double red[];  // [kGfWaves - 1][NTT][256]
  if (wave > 0) {
    double* rw = red + (size_t)(wave - 1) * NTT * 256;
#pragma unroll
    for (int x = 0; x < NTT; ++x)
#pragma unroll
      for (int r = 0; r < 4; ++r) rw[x * 256 + r * 64 + lane] = acc[x][r];
  }
  __syncthreads();
  if (wave == 0) {
    double* dst = part + (((size_t)b * K + k) * C + c) * (size_t)NTT * 256;
#pragma unroll
    for (int x = 0; x < NTT; ++x)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        double t = acc[x][r];
#pragma unroll
        for (int w = 0; w < kGfWaves - 1; ++w) t += red[((size_t)w * NTT + x) * 256 + r * 64 + lane];
        dst[(size_t)x * 256 + r * 64 + lane] = t;
      }
  }
}

// ------------------------------------------------------------------ small dense helpers (LDS)
// Cholesky AND the inverse of its factor in one elimination.  With A = L_u D L_u^T (unit lower
// L_u, pivots D) the Cholesky factor is L = L_u D^1/2 and X = L^-1 = D^-1/2 L_u^-1; eliminating
// on [A | I] turns the identity into M = L_u^-1 with the SAME multipliers as the trailing update.
// Thread t owns the lower-triangle positions t, t + 256, ... (row-major) of BOTH triangles in
// registers: `val` (trailing matrix, live while j < c) and `mv` (M, live while c <= j < r), so
// at step j a position does exactly one update  x -= (a_rj / d_j) * other  with
// other = a_cj (pivot column, trailing part) or m_jc (row j of M, final since step j - 1).
// LDS only carries what other threads read: the pivot columns of A (published when the column
// becomes the next pivot column) and the rows of M (published when the row becomes final) --
// one barrier per step, every load unconditional (clamped index, masked use) so the loads of a
// step are issued back to back.  Round 1 ran Cholesky (23 us for E = 40), then a forward
// substitution with one thread per column (30 us) behind it.
// On return: a[j][j] = pivots d_j, x[r][c] (c <= r) = X = L^-1.  Returns 0 or 1 + the index of
// the first non-positive pivot (LAPACK dpotrf INFO), uniform over the workgroup.
template <int Q>  // owned positions per thread: Q * kGfThreads >= E (E + 1) / 2
__device__ int gf_chol_inverse_q(double* a, double* x, int E, int ld, int tid, int* info_sm) {
  const int ntri = E * (E + 1) / 2;
  double* pinv = x + 1;  // 1 / d_j in x[0][1 + j]: row 0 of the (lower-triangular) result is x[0][0]
  int er[Q], ec[Q];
  double val[Q], mv[Q];
  auto row_of = [](int idx) {
    int r = (int)((sqrt(8.0 * idx + 1.0) - 1.0) * 0.5);
    while (r * (r + 1) / 2 > idx) --r;
    while ((r + 1) * (r + 2) / 2 <= idx) ++r;
    return r;
  };
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int idx = tid + q * kGfThreads;
    const int r = row_of(idx);
    const bool in = idx < ntri;
    er[q] = in ? r : 0;            // padding slots sit on (0, 0): never updated (no j < 0)
    ec[q] = in ? idx - r * (r + 1) / 2 : 0;
    val[q] = in ? a[r * ld + ec[q]] : 0.0;
    mv[q] = (er[q] == ec[q]) ? 1.0 : 0.0;
    if (in && er[q] == 0) {
      x[0] = 1.0;  // row 0 of M is final from the start
      const bool bad = !(val[q] > 0.0) || !(val[q] < 1.79e308);
      *info_sm = bad ? 1 : 0;
      pinv[0] = bad ? 1.0 : 1.0 / val[q];
    }
  }
  __syncthreads();  // everyone holds its elements; column 0 of `a` is the first pivot column
  for (int j = 0; j < E; ++j) {
    const double inv_d = pinv[j];
    double arj[Q], oth[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      arj[q] = a[er[q] * ld + j];
      // j < c: a_cj of the pivot column; otherwise m_jc (zero-filled above the diagonal of M
      // is never read: c <= j there)
      oth[q] = (j < ec[q]) ? a[ec[q] * ld + j] : x[j * ld + ec[q]];
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const double upd = (arj[q] * inv_d) * oth[q];
      const bool trailing = j < ec[q];               // implies j < r
      const bool inverse = !trailing && j < er[q];   // c <= j < r
      val[q] = trailing ? val[q] - upd : val[q];
      mv[q] = inverse ? mv[q] - upd : mv[q];
      if (trailing && ec[q] == j + 1) {
        a[er[q] * ld + j + 1] = val[q];  // next pivot column
        if (er[q] == j + 1) {
          // ... and its pivot's reciprocal, by the one thread that owns it: the division
          // overlaps this wavefront's other positions instead of heading everybody's next step
          const bool bad = !(val[q] > 0.0) || !(val[q] < 1.79e308);
          if (bad && *info_sm == 0) *info_sm = j + 2;
          pinv[j + 1] = bad ? 1.0 : 1.0 / val[q];
        }
      }
      if (er[q] == j + 1) x[er[q] * ld + ec[q]] = mv[q];  // row j + 1 of M is final
    }
    __syncthreads();
  }
  // X = D^-1/2 M
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const double dj = a[er[q] * ld + er[q]];
    const double piv = sqrt((dj > 0.0 && dj < 1.79e308) ? dj : 1.0);
    mv[q] = mv[q] / piv;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < Q; ++q)
    if (tid + q * kGfThreads < ntri) x[er[q] * ld + ec[q]] = mv[q];
  __syncthreads();
  return *info_sm;
}

// every masked-off position still costs its instruction slots, so the elimination is compiled
// for the number of positions a thread really owns (E = 40: 4 instead of the 9 of E = 63)
__device__ int gf_chol_inverse(double* a, double* x, int E, int ld, int tid, int* info_sm) {
  const int q = (E * (E + 1) / 2 + kGfThreads - 1) / kGfThreads;
  switch (q) {
    case 1: return gf_chol_inverse_q<1>(a, x, E, ld, tid, info_sm);
    case 2: return gf_chol_inverse_q<2>(a, x, E, ld, tid, info_sm);
    case 3: return gf_chol_inverse_q<3>(a, x, E, ld, tid, info_sm);
    case 4: return gf_chol_inverse_q<4>(a, x, E, ld, tid, info_sm);
    case 5: return gf_chol_inverse_q<5>(a, x, E, ld, tid, info_sm);
    case 6: return gf_chol_inverse_q<6>(a, x, E, ld, tid, info_sm);
    case 7: return gf_chol_inverse_q<7>(a, x, E, ld, tid, info_sm);
    default: return gf_chol_inverse_q<8>(a, x, E, ld, tid, info_sm);  // E <= 63
  }
}

// cov (E x E, global) -> P = L^-T (global, E x E, upper triangular; parameter `out_mq`) and
// offset = -E/2 ln 2pi - sum ln L_dd;
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
#ifdef PBBSS_GF_STAMP
  long long g0 = wall_clock64();
#endif
  const int info = gf_chol_inverse(l, x, E, ld, tid, info_sm);
#ifdef PBBSS_GF_STAMP
  long long g1 = wall_clock64(), g2 = g1;
#endif
  // the E-step consumes the whitening matrix itself: P = X^T = L^-T (upper triangular; the
  // reference's precision_cholesky, gaussian.py:26-30), q = |P d|^2
  for (int idx = tid; idx < E * E; idx += kGfThreads) {
    const int r = idx / E, cc = idx % E;
    out_mq[(size_t)r * E + cc] = (cc >= r) ? x[cc * ld + r] : 0.0;
  }
#ifdef PBBSS_GF_STAMP
  __syncthreads();
  long long g3 = wall_clock64();
  if (tid == 0 && blockIdx.x == 0 && blockIdx.y == 0)
    printf("gf_factor: cholesky %lld tri_inverse %lld xxt %lld (10 ns)\n", g1 - g0, g2 - g1, g3 - g2);
#endif
  if (tid < kWave) {  // E <= 63: one wavefront sums the log-pivots
    double sl = (tid < E) ? 0.5 * log(l[tid * ld + tid]) : 0.0;  // ln L_dd = ln sqrt(d)
    sl = wave_sum(sl);
    if (tid == 0) *out_offset = -0.5 * E * kLn2PiGf - sl;  // sum_d ln P_dd = -sum_d ln L_dd
  }
  __syncthreads();
  return info;
}

// ------------------------------------------------------------------ finalize
// Stage 1, grid (tile, K, B): ordered sum of the NP wave partials of one tile, one element per
// thread, eight loads in flight.  (One workgroup per class walking all partials of all tiles
// was a 250 us chain of L2 round trips.)
__global__ void __launch_bounds__(kGfThreads)
    gf_reduce_kernel(const double* __restrict__ part, int NP, int NTT, int K,
                     double* __restrict__ gsum) {
  const int x = blockIdx.x, k = blockIdx.y;
  const int64_t b = blockIdx.z;
  const int e = threadIdx.x;
  const double* p = part + (((size_t)b * K + k) * NP) * (size_t)NTT * 256 + (size_t)x * 256 + e;
  const size_t stride = (size_t)NTT * 256;
  double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int q = 0;
  for (; q + 7 < NP; q += 8) {
    double a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = p[(size_t)(q + u) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] += a[u];
  }
  for (; q < NP; ++q) t[0] += p[(size_t)q * stride];
  gsum[(((size_t)b * K + k) * NTT + x) * 256 + e] =
      ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
}

// Stage 2, one workgroup per (b, k): G -> mean, covariance (gaussian.py:155-190), optionally the
// factorisation.
template <int NT, typename TS>
__global__ void __launch_bounds__(kGfThreads)
    gf_finalize_kernel(const double* __restrict__ gsum, const TS* __restrict__ y, int64_t N,
                       int E, int K, double* __restrict__ out_mean, double* __restrict__ out_cov,
                       double* out_mq, double* out_offset, double* out_s0,
                       int32_t* out_status, const double* __restrict__ gshift) {
  constexpr int NTT = NT * (NT + 1) / 2;
  constexpr int P = 16 * NT;
  extern __shared__ double sm[];
  double* G = sm;  // [P][P + 1]
  __shared__ int info_sm;
  const int tid = threadIdx.x;
  const int k = blockIdx.x;
  const int64_t b = blockIdx.y;
#ifdef PBBSS_GF_STAMP
  long long h0 = wall_clock64();
#endif
  const double* pb = gsum + ((size_t)b * K + k) * (size_t)NTT * 256;
  // element e of a tile: e = r * 64 + l  ->  row 4 r + l / 16, column l % 16
  for (int idx = tid; idx < NTT * 256; idx += kGfThreads) {
    const double tot = pb[idx];
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
  for (int d = tid; d < E; d += kGfThreads)
    mean[d] = (gshift ? gshift[d] : (double)yb[d]) + G[E * (P + 1) + d] / den;
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
#ifdef PBBSS_GF_STAMP
  if (tid == 0 && k == 0 && b == 0) printf("gf_finalize: gather+cov %lld (10 ns)\n", wall_clock64() - h0);
#endif
  if (out_mq) {
    __threadfence_block();
    int info = gf_factor(cov, E, sm, &info_sm, out_mq + ((size_t)b * K + k) * (size_t)E * E,
                         out_offset + (size_t)b * K + k, tid);
    if (info && tid == 0 && out_status) atomicOr(out_status, (int32_t)PBBSS_ST_NOT_POSDEF);
  }
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
// q_n = |P_k (y_n - mu_k)|^2 with the upper-triangular whitening matrix P_k = L_k^-T that the
// factorisation leaves behind (the reference's precision_cholesky, gaussian.py:26-56), on the
// FP64 matrix pipe: per 16 samples and class the product Z = [P_k | -P_k (mu_k - c)] [y - c; 1]
// as 16x16x4 tiles, zero tiles below the diagonal skipped (24 instead of 36 MFMA at P = 48).
//   * B operand = the samples.  Lane (i = l % 16, g = l / 16) needs y[n0 + i][kk + g] for every
//     chunk kk: it loads those NK values straight from global memory into registers (a sample
//     row is consumed completely by the NK loads of a lane quartet, so every fetched line is
//     used), subtracts the common shift c = mean of class 0 (keeps the augmented form free of
//     cancellation) and keeps them for ALL classes and row tiles -- no LDS staging of the
//     samples, no barrier inside the sample loop, the loads of the next group are issued before
//     the MFMAs of the current one.  Round 1 staged the centred block per class through LDS:
//     K global reads of y and two barriers per 64 samples.
//   * A operand = [P_k | -P_k (mu_k - c)], `KC` classes at a time staged in LDS ONCE per
//     workgroup (S = 64 * groups samples).
// The log-pdfs of a workgroup's samples are parked in LDS ([K][S]) for the softmax at the end.
// grid (ceil(N / S), B).
template <int NT, typename TS>
__global__ void __launch_bounds__(kGfThreads)
    gf_logpdf_kernel(const TS* __restrict__ y, int64_t N, int E, int K, int KC, int groups,
                     const double* __restrict__ mean, const double* __restrict__ mq,
                     const double* __restrict__ offset, const double* __restrict__ weight,
                     double* __restrict__ out_lp, double* __restrict__ out_aff) {
  constexpr int P = 16 * NT;
  constexpr int NK = P / 4;
  extern __shared__ double sm[];
  double* A = sm;                                 // [KC][P][P + 1]
  double* lp = sm + (size_t)KC * P * (P + 1);     // [K][S]
  double* dmu = lp + (size_t)K * 64 * groups;     // [KC][P]
  double* colE = dmu + (size_t)KC * P;            // [KC][P]
  double* offs = colE + (size_t)KC * P;           // [KC]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t b = blockIdx.y;
  const int S = 64 * groups;
  const int64_t nb0 = (int64_t)blockIdx.x * S;
  const TS* yb = y + (size_t)b * N * E;
  const int i = lane & 15, g = lane >> 4;
  const double* cm = mean + (size_t)b * K * E;  // the shift: mean of class 0
  double creg[NK];
  int dcl[NK];
#pragma unroll
  for (int c = 0; c < NK; ++c) {
    const int d = 4 * c + g;
    dcl[c] = (d < E) ? d : 0;
    creg[c] = cm[dcl[c]];
  }
  auto fetch = [&](int gi, TS (&raw)[NK]) {
    const int64_t n = nb0 + 16 * (gi * kGfWaves + wave) + i;
    const TS* row = yb + (size_t)((n < N) ? n : N - 1) * E;
#pragma unroll
    for (int c = 0; c < NK; ++c) raw[c] = row[dcl[c]];
  };
  for (int k0 = 0; k0 < K; k0 += KC) {
    const int kcn = (K - k0 < KC) ? K - k0 : KC;
    __syncthreads();  // the previous chunk's MFMA reads of A are done
    // every load of the staging below is independent and issued in batches: a plain strided
    // loop is one L2 round trip per trip (27 of them for three 48 x 48 matrices)
    for (int kc = 0; kc < kcn; ++kc) {
      const double* mk = mq + ((size_t)b * K + k0 + kc) * (size_t)E * E;
      constexpr int R = P * P / kGfThreads;
      double rawa[R];
#pragma unroll
      for (int q = 0; q < R; ++q) {
        const int idx = tid + q * kGfThreads;
        const int r = idx / P, cc = idx % P;
        rawa[q] = mk[(size_t)((r < E) ? r : 0) * E + ((cc < E) ? cc : 0)];
      }
#pragma unroll
      for (int q = 0; q < R; ++q) {
        const int idx = tid + q * kGfThreads;
        const int r = idx / P, cc = idx % P;
        A[((size_t)kc * P + r) * (P + 1) + cc] = (r < E && cc < E) ? rawa[q] : 0.0;
      }
    }
    for (int idx = tid; idx < kcn * E; idx += kGfThreads) {  // mu_k - c
      const int kc = idx / E, cc = idx % E;
      dmu[kc * P + cc] = mean[((size_t)b * K + k0 + kc) * E + cc] - cm[cc];
    }
    if (tid < kcn) offs[tid] = offset[(size_t)b * K + k0 + tid];
    __syncthreads();
    for (int idx = tid; idx < kcn * E; idx += kGfThreads) {  // column E: -P_k (mu_k - c)
      const int kc = idx / E, r = idx % E;
      const double* ar = A + ((size_t)kc * P + r) * (P + 1);
      const double* dm = dmu + kc * P;
      double acc = 0.0;
      for (int c0 = r; c0 < E; c0 += 8) {
        double av[8], dv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int cc = (c0 + u < E) ? c0 + u : r;
          av[u] = ar[cc];
          dv[u] = dm[cc];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += (c0 + u < E) ? av[u] * dv[u] : 0.0;
      }
      colE[idx] = -acc;
    }
    __syncthreads();  // column E is written only after every product above has read its row
    for (int idx = tid; idx < kcn * E; idx += kGfThreads)
      A[((size_t)(idx / E) * P + idx % E) * (P + 1) + E] = colE[idx];
    __syncthreads();
    TS raw[NK];
    fetch(0, raw);
    for (int gi = 0; gi < groups; ++gi) {
      const int64_t n0 = nb0 + 16 * (gi * kGfWaves + wave);
      if (n0 >= N) break;  // wave-uniform
      double bv[NK];
#pragma unroll
      for (int c = 0; c < NK; ++c) {
        const int d = 4 * c + g;
        bv[c] = (d < E) ? (double)raw[c] - creg[c] : ((d == E) ? 1.0 : 0.0);
      }
      if (gi + 1 < groups) fetch(gi + 1, raw);  // in flight during this group's MFMAs
      for (int kc = 0; kc < kcn; ++kc) {
        const double* ak = A + (size_t)kc * P * (P + 1);
        // all A operands of the class first (one batch of LDS reads, nothing waits inside the
        // MFMA sequence), then the NT row tiles as independent accumulator chains interleaved
        // chunk by chunk: consecutive MFMAs never depend on each other
        double av[NT][NK];
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
          for (int c = 4 * ti; c < NK; ++c)  // upper triangular: column chunks >= the row tile
            av[ti][c] = ak[(16 * ti + i) * (P + 1) + 4 * c + g];  // A[i][k = g]
        double4_t z[NT];
#pragma unroll
        for (int ti = 0; ti < NT; ++ti) z[ti] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int c = 0; c < NK; ++c)
#pragma unroll
          for (int ti = 0; ti < NT; ++ti)
            if (c >= 4 * ti)
              z[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ti][c], bv[c], z[ti], 0, 0, 0);
        // z[ti][r] = (P_k (y - mu_k))[out dim 16 ti + 4 r + g] of sample i
        double qpart = 0.0;
#pragma unroll
        for (int ti = 0; ti < NT; ++ti)
#pragma unroll
          for (int r = 0; r < 4; ++r) qpart = fma(z[ti][r], z[ti][r], qpart);
        qpart += __shfl_xor(qpart, 16, 64);
        qpart += __shfl_xor(qpart, 32, 64);
        if (g == 0)
          lp[(size_t)(k0 + kc) * S + 16 * (gi * kGfWaves + wave) + i] =
              offs[kc] - 0.5 * qpart;
      }
    }
  }
  __syncthreads();
  for (int sidx = tid; sidx < S; sidx += kGfThreads) {
    const int64_t n = nb0 + sidx;
    if (n >= N) continue;
    double mx = -1.79e308;
    for (int k = 0; k < K; ++k) mx = fmax(mx, lp[k * S + sidx]);
    if (out_lp)
      for (int k = 0; k < K; ++k) out_lp[((size_t)b * K + k) * N + n] = lp[k * S + sidx];
    if (out_aff) {  // mixture_model_utils.py:30-47, affiliation_eps = 0
      double den = 0.0;
      for (int k = 0; k < K; ++k) den += exp(lp[k * S + sidx] - mx) * weight[(size_t)b * K + k];
      den = fmax(den, kTiny);
      for (int k = 0; k < K; ++k)
        out_aff[((size_t)b * K + k) * N + n] =
            exp(lp[k * S + sidx] - mx) * weight[(size_t)b * K + k] / den;
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
  // two workgroups per CU (one wave's loads and operand arithmetic run under the other's MFMAs:
  // 75 -> 68 us for the scatter at N = 256 500, K = 3; three or four are slower again), at
  // least 64 samples per wave.  PBBSS_GF_CHUNK_MUL overrides the factor for A/B runs.
  static const int mul = [] {
    const char* v = getenv("PBBSS_GF_CHUNK_MUL");
    return (v && atoi(v) > 0) ? atoi(v) : 2;
  }();
  int64_t c = 256 * mul / (B * K < 256 ? B * K : 256);
  const int64_t maxc = (N + 4 * 64 - 1) / (4 * 64);
  if (c > maxc) c = maxc;
  return (int)(c < 1 ? 1 : c);
}

template <int NT, typename TS>
int gf_fit_go(const void* y, int64_t B, int64_t N, int E, int K, const double* w,
              const double* sal, double* part, double* out_mean, double* out_cov, double* out_mq,
              double* out_offset, double* out_s0, int32_t* out_status, hipStream_t s,
              const double* gshift, const PartialReduce* reduce) {
  constexpr int P = 16 * NT;
  const int C = gf_chunks(B, K, N);
  int64_t Lw = (N + (int64_t)C * kGfWaves - 1) / ((int64_t)C * kGfWaves);
  Lw = (Lw + 3) / 4 * 4;
  constexpr int NTT = NT * (NT + 1) / 2;
  hipLaunchKernelGGL((gf_scatter_kernel<NT, TS>), dim3((unsigned)C, (unsigned)K, (unsigned)B),
                     dim3(kGfThreads), (kGfWaves - 1) * NTT * 256 * sizeof(double), s,
                     static_cast<const TS*>(y), N, E, K, w, sal, C, Lw, part, gshift);
  size_t ldsd = (size_t)P * (P + 1);
  if (out_mq && 2 * (size_t)E * (E + 1) > ldsd) ldsd = 2 * (size_t)E * (E + 1);
  const size_t lds = ldsd * sizeof(double);
  auto kfn = gf_finalize_kernel<NT, TS>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return PBBSS_ERR_HIP;
  double* gsum = part + (size_t)B * K * C * NTT * 256;  // behind the workgroup partials
  hipLaunchKernelGGL(gf_reduce_kernel, dim3((unsigned)NTT, (unsigned)K, (unsigned)B),
                     dim3(kGfThreads), 0, s, part, C, NTT, K, gsum);
  if (reduce) {  // bins sharded over ranks: the reduced Gram tiles of all ranks (common shift)
    if (int rc = reduce->fn(reduce->ctx, gsum, (size_t)B * K * NTT * 256, s); rc != PBBSS_OK) return rc;
  }
  hipLaunchKernelGGL(kfn, dim3((unsigned)K, (unsigned)B), dim3(kGfThreads), lds, s, gsum,
                     static_cast<const TS*>(y), N, E, K, out_mean, out_cov, out_mq, out_offset,
                     out_s0, out_status, gshift);
  return gf_ok();
}

template <int NT, typename TS>
int gf_logpdf_go(const void* y, int64_t B, int64_t N, int E, int K, const double* mean,
                 const double* mq, const double* offset, const double* weight, double* out_lp,
                 double* out_aff, hipStream_t s) {
  constexpr int P = 16 * NT;
  // classes staged per pass: whitening matrices within 64 KiB of LDS
  int KC = (int)((64 * 1024) / (sizeof(double) * P * (P + 1)));
  if (KC > K) KC = K;
  if (KC < 1) KC = 1;
  // samples per workgroup: 512 (two workgroups per CU cover 256 500 samples in one round) unless
  // the parked log-pdfs [K][S] would exceed 32 KiB
  int groups = (int)(32 * 1024 / (sizeof(double) * 64 * (size_t)K));
  if (groups > 8) groups = 8;
  if (groups < 1) groups = 1;
  const size_t lds =
      ((size_t)KC * P * (P + 1) + (size_t)K * 64 * groups + (size_t)KC * (2 * P + 1)) *
      sizeof(double);
  auto kfn = gf_logpdf_kernel<NT, TS>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return PBBSS_ERR_HIP;
  const int64_t per = 64 * (int64_t)groups;
  hipLaunchKernelGGL(kfn, dim3((unsigned)((N + per - 1) / per), (unsigned)B), dim3(kGfThreads),
                     lds, s, static_cast<const TS*>(y), N, E, K, KC, groups, mean, mq, offset,
                     weight, out_lp, out_aff);
  return gf_ok();
}

}  // namespace

size_t gauss_full_partial_doubles(int64_t B, int64_t N, int E, int K) {
  const int NT = (E + 1 + 15) / 16;
  const int C = gf_chunks(B, K, N);
  return (size_t)B * K * (C + 1) * (size_t)(NT * (NT + 1) / 2) * 256;  // + tile sums
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
                          double* out_s0, int32_t* out_status, hipStream_t s,
                          const double* shift, const PartialReduce* reduce) {
  if (E < 1 || E > kGaussFullMaxE || K < 1 || B < 1 || B > 65535 || K > 65535)
    return PBBSS_ERR_UNSUPPORTED;
  if ((shift || reduce) && B != 1) return PBBSS_ERR_INVALID_ARG;  // one mixture over the ranks
  PBBSS_GF_DISPATCH(gf_fit_go, y, B, N, E, K, weights, sal, part, out_mean, out_cov, out_mq,
                    out_offset, out_s0, out_status, s, shift, reduce)
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
