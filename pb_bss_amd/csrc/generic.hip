// Generic-size path of the cACGMM trainer for 9 <= D <= 32 sensors (the persistent kernel of
// cacgmm_em.hpp is specialised for D <= 8, where a D x D matrix maps onto one wavefront).
// Same functions as SURVEY.md section 8a rows a2-a8 / a10, decomposed into kernels that the
// C ABI enqueues back to back (no host synchronisation); per EM iteration:
//   gen_cov    one workgroup per bin: the upper triangle of C_k = sum_t w_kt y_t y_t^H in 4x2
//              register tiles, frames staged through a 64-frame LDS tile and split over
//              frame groups; M-step or PSD normalisation, class weights
//                                                           (cacg.py:253-342, beamformer.py:59)
//   gen_inv    one workgroup per matrix: Gauss-Jordan inverse + log det in LDS, accepted only
//              when a condition bound proves that no eigenvalue would be floored (fast path
//              between iterations, the class log-pdf being invariant to the scale of B)
//   gen_heev   one workgroup per matrix, skipped where gen_inv succeeded and not the last
//              iteration: parallel cyclic Jacobi in LDS (D/2 disjoint rotations per round,
//              ping-pong buffers), ascending eigenvalues, the reference's normalisation and
//              floor                                        (cacg.py:82-132)
//   gen_eig_to_inv  (V, lambda) -> B^-1 = V diag(1/lambda) V^H, log det for the other matrices
//   gen_estep  grid (bins, frames / 256), thread = frame: q = y^H B^-1 y over the upper
//              triangle with B^-1 fetched by scalar loads (SGPR operands, no LDS), log-domain
//              softmax                                     (cacg.py:167-203, mm_utils.py:7-55)
// cov / inv / heev pad matrices to DP = 16 or 32, the E-step to multiples of 4 (templates) so
// that register arrays index statically.  Float64 arithmetic throughout, as the D <= 8 path.
#include "generic.hpp"
#include <cmath>
#include "generic_dev.hpp"
#include <cstdlib>
#include <type_traits>

namespace pbbss {
namespace {

constexpr int kGenMaxK = 19;   // classes of the generic path (the reference asserts K < 20, cacgmm.py:249)
constexpr int kCovMaxK = 6;    // classes accumulated by one gen_cov launch (register tiles)
constexpr int kTile = 64;  // frames per LDS tile of gen_cov

// doubles of gen_cov's staging area: the frame tile, later the per-thread partial tiles
__host__ __device__ constexpr size_t gen_cov_stage_doubles(int DP) {
  const size_t frames = (size_t)kTile * (DP + 1) * 2, parts = (size_t)kGenThreads * 8 * 2;
  return frames > parts ? frames : parts;
}

__device__ __forceinline__ double block_sum(double v, double* red, int tid) {
  return gen_block_sum(v, red, tid);
}

template <typename YS>
__device__ __forceinline__ void load_y(const void* y, int layout, int64_t b, int t, int d, int T,
                                       int D, double& re, double& im) {
  const size_t idx = (layout == PBBSS_LAYOUT_TD) ? ((size_t)b * T + t) * D + d
                                                 : ((size_t)b * D + d) * T + t;
  const YS* p = static_cast<const YS*>(y) + 2 * idx;
  re = (double)p[0];
  im = (double)p[1];
}

// ------------------------------------------------------------------ E-step
// The model reaches the E-step as B_k^-1 and log det B_k in the padded "inverse state"
// ((N, LD, LD) complex128, LD = gen_state_ld(D), zeros beyond D).  The entries are uniform
// over a workgroup, so they are fetched with scalar loads (constant address space) and enter
// the multiply-accumulates as SGPR operands: no LDS traffic, and the vector ALU is the bound.
struct GenEstep {
  const void* y;
  int layout;
  int64_t B;
  int T, D, K;
  const double* inv;     // c128 (B*K, LD, LD)
  const double* logdet;  // (B*K)
  const double* weight;
  int64_t wb, wk, wt;
  const uint8_t* activity;
  double eps;
  double* out_aff;
  double* out_q;
  double* out_logpdf;
  // EM loop only: the M-step weight gamma sal / max(q, 10 tiny) / |y|^2 of every frame
  // (cacg.py:310, :322) for gen_cov2, and the all-zero-frame flag of the bin
  const double* saliency;  // (B,T) or null
  double* out_mweight;     // (B,K,T) or null
  int32_t* out_zero;       // (B) or null: set to 1 where a frame is all-zero
  int raw;                 // 1: the observation is raw (unit-normalised here) whatever its layout
  // joint spatial + spectral models (gcacgmm.py:66-117): the posterior's exponent is
  // spatial_scale * log-pdf + extra[b,k,t]; out_logpdf / out_q stay the spatial quantities
  const double* extra;     // (B,K,T) or null
  double spatial_scale;
};

// static group stream of the E-step operands: row i (diagonal .. DP - 1) has ceil((DP - i) / 8)
// groups of eight complex entries
template <int DP>
constexpr int gen_e_groups_before(int row) {
  int n = 0;
  for (int i = 0; i < row; ++i) n += (DP - i + 7) / 8;
  return n;
}
template <int DP>
constexpr int gen_e_row_of(int n) {
  int i = 0;
  while (n >= (DP - i + 7) / 8) {
    n -= (DP - i + 7) / 8;
    ++i;
  }
  return i;
}

template <int DP, typename YS>
__global__ void __launch_bounds__(kGenThreads) gen_estep_kernel(GenEstep a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* qs = reinterpret_cast<double*>(smem);   // [K][thread] quadratic forms (private slots)
  double* ls = qs + (size_t)a.K * kGenThreads;    // [K][thread] log-pdf, then unnormalised posterior
  const int tid = threadIdx.x;
  const int64_t b = blockIdx.x;
  const int D = a.D, K = a.K, T = a.T;
  const int t = blockIdx.y * kGenThreads + tid;  // thread = frame
  const bool valid = t < T;
  double yr[DP], yi[DP], n2 = 0.0;
  {
    // unconditional loads at clamped indices into raw registers, masks afterwards (a guarded
    // load that is converted inside its guard compiles to branch + load + s_waitcnt vmcnt(0):
    // D serial round trips).  (B, T, D) input is read with a lane stride of D
    // complex numbers; staging it through LDS in coalesced slabs was measured and is slower
    // (105 vs 101 us at D = 29, 75 vs 66 us at D = 24: the lines are L2 hits either way).
    const int tc = valid ? t : 0;
    YS rr[DP], ri[DP];
    const size_t st = (a.layout == PBBSS_LAYOUT_TD) ? (size_t)D : 1;
    const size_t sd = (a.layout == PBBSS_LAYOUT_TD) ? 1 : (size_t)T;
    const YS* base = static_cast<const YS*>(a.y) + 2 * ((size_t)b * T * D + (size_t)tc * st);
#pragma unroll
    for (int d = 0; d < DP; ++d) {
      const YS* p = base + 2 * ((size_t)((d < D) ? d : 0) * sd);
      rr[d] = p[0];
      ri[d] = p[1];
    }
#pragma unroll
    for (int d = 0; d < DP; ++d) {
      const bool ok = d < D && valid;
      yr[d] = ok ? (double)rr[d] : 0.0;
      yi[d] = ok ? (double)ri[d] : 0.0;
      n2 += yr[d] * yr[d] + yi[d] * yi[d];
    }
  }
  // raw observations are unit-normalised (zero frames stay zero, utils.py:223-256)
  const double inv = a.raw ? ((n2 > 0.0) ? 1.0 / n2 : 0.0) : 1.0;
  // y^H A y = sum_i A_ii |y_i|^2 + 2 Re sum_i conj(y_i) sum_{j>i} A_ij y_j  (A Hermitian).
  // Operand feed: row i of A_k from the diagonal on is contiguous in the state and is cut into
  // groups of eight complex entries; lane l loads double (l & 15) of a group, so one VGPR pair
  // carries the group in each of the four rows of 16 lanes and the FMAs pick their operand
  // with row_newbcast (fmac_row_bcast): one coalesced 128-byte load per 32 FMAs instead of a
  // scalar load or an LDS broadcast per operand.  The groups of a class form one static stream
  // (rows in order), fetched kAhead groups in front of the FMAs through a register ring; the
  // scheduling barrier keeps hipcc from sinking the loads back to their first use.  No bounds
  // tests: A and y are zero beyond D, the lane index is clamped to the matrix.
  const int l16 = tid & 15;
  constexpr int NGT = gen_e_groups_before<DP>(DP);  // groups of one class
  constexpr int kAhead = (NGT < 6) ? NGT : 6, kRing = kAhead + 1;
  double ring[kRing];
  auto load_group = [&](const double* A, auto nc) {
    constexpr int n = nc;
    constexpr int i = gen_e_row_of<DP>(n), g = n - gen_e_groups_before<DP>(i);
    const int idx = 2 * (i * DP + i) + 16 * g + l16;
    ring[n % kRing] = A[idx < 2 * DP * DP ? idx : 2 * DP * DP - 1];
  };
  {
    const double* A0 = a.inv + (size_t)b * K * DP * DP * 2;
    static_for<0, kAhead>([&](auto nc) { load_group(A0, nc); });
  }
#pragma unroll 1
  for (int k = 0; k < K; ++k) {
    const double* A = a.inv + ((size_t)b * K + k) * DP * DP * 2;
    double q = 0.0;
    double ura = 0.0, urb = 0.0, uia = 0.0, uib = 0.0;  // Re u = ura - urb, Im u = uia + uib
    static_for<0, NGT>([&](auto nc) {
      constexpr int n = nc;
      constexpr int i = gen_e_row_of<DP>(n), g = n - gen_e_groups_before<DP>(i);
      constexpr int ng = (DP - i + 7) / 8;
      if constexpr (n + kAhead < NGT) load_group(A, std::integral_constant<int, n + kAhead>{});
      __builtin_amdgcn_sched_barrier(0);
      const double grp = ring[n % kRing];
      if constexpr (g == 0) {
        ura = urb = uia = uib = 0.0;
        fmac_row_bcast<0>(q, grp, fma(yr[i], yr[i], yi[i] * yi[i]));
      }
      static_for<0, 8>([&](auto cc) {
        constexpr int c = cc, j = i + 8 * g + c;
        if constexpr (j > i && j < DP) {
          fmac_row_bcast<2 * c>(ura, grp, yr[j]);
          fmac_row_bcast<2 * c + 1>(urb, grp, yi[j]);
          fmac_row_bcast<2 * c>(uia, grp, yi[j]);
          fmac_row_bcast<2 * c + 1>(uib, grp, yr[j]);
        }
      });
      if constexpr (g == ng - 1) q = fma(2.0, fma(yr[i], ura - urb, yi[i] * (uia + uib)), q);
    });
    __builtin_amdgcn_sched_barrier(0);
    {  // first groups of the next class (the last class re-reads its own: harmless)
      const double* An = a.inv + ((size_t)b * K + (k + 1 < K ? k + 1 : k)) * DP * DP * 2;
      static_for<0, kAhead>([&](auto nc) { load_group(An, nc); });
    }
    qs[k * kGenThreads + tid] = q;
  }
  if (!valid) return;
  // softmax over the classes in three rolled passes through the thread's LDS slots (K <= 19
  // without per-class register arrays)
  double mx = -1.79e308;
#pragma unroll 1
  for (int k = 0; k < K; ++k) {
    const double q = fmax(fabs(qs[k * kGenThreads + tid] * inv), kTiny);  // cacg.py:185-199
    const double lp = -(double)D * log(q) - a.logdet[b * K + k];          // cacg.py:151
    qs[k * kGenThreads + tid] = q;
    if (a.out_logpdf) a.out_logpdf[((size_t)b * K + k) * T + t] = lp;
    const double le = a.extra ? fma(a.spatial_scale, lp, a.extra[((size_t)b * K + k) * T + t]) : lp;
    ls[k * kGenThreads + tid] = le;
    mx = fmax(mx, le);
    if (a.out_q) a.out_q[((size_t)b * K + k) * T + t] = q;
  }
  if (!a.out_aff) return;
  double den = 0.0;
#pragma unroll 1
  for (int k = 0; k < K; ++k) {
    double v = exp(ls[k * kGenThreads + tid] - mx) * a.weight[b * a.wb + k * a.wk + (int64_t)t * a.wt];
    if (a.activity) v *= (double)a.activity[((size_t)b * K + k) * T + t];
    ls[k * kGenThreads + tid] = v;
    den += v;
  }
  den = fmax(den, kTiny);
#pragma unroll 1
  for (int k = 0; k < K; ++k) {
    double gam = ls[k * kGenThreads + tid] / den;
    if (a.eps != 0.0) gam = fmin(fmax(gam, a.eps), 1.0 - a.eps);
    a.out_aff[((size_t)b * K + k) * T + t] = gam;
    if (a.out_mweight) {
      const double sal = a.saliency ? a.saliency[(size_t)b * T + t] : 1.0;
      a.out_mweight[((size_t)b * K + k) * T + t] =
          gam * sal / fmax(qs[k * kGenThreads + tid], 10.0 * kTiny) * inv;
    }
  }
  if (a.out_zero && a.raw && !(n2 > 0.0)) a.out_zero[b] = 1;
}

// (V, lambda) -> inverse state for the matrices not already marked ok (cacg.py:167-183 forms
// the same product V diag(1/lambda) V^H before the quadratic form)
struct GenEigToInv {
  const double* eigvec;  // c128 (N,D,D)
  const double* eigval;  // (N,D)
  int D, LD;
  const int32_t* ok;     // (N) or null
  double* inv;           // c128 (N,LD,LD)
  double* logdet;        // (N)
};

__global__ void __launch_bounds__(kGenThreads) gen_eig_to_inv_kernel(GenEigToInv g) {
  const int64_t n = blockIdx.x;
  if (g.ok && g.ok[n]) return;
  const int tid = threadIdx.x, D = g.D, LD = g.LD;
  const double* v = g.eigvec + (size_t)n * D * D * 2;
  const double* lam = g.eigval + (size_t)n * D;
  for (int e = tid; e < LD * LD; e += kGenThreads) {
    const int i = e / LD, j = e - i * LD;
    double sr = 0.0, si = 0.0;
    if (i < D && j < D) {
      for (int m = 0; m < D; ++m) {
        const double il = 1.0 / lam[m];
        const double ar = v[(i * D + m) * 2], ai = v[(i * D + m) * 2 + 1];
        const double br = v[(j * D + m) * 2], bi = v[(j * D + m) * 2 + 1];
        sr += (ar * br + ai * bi) * il;   // V_im conj(V_jm)
        si += (ai * br - ar * bi) * il;
      }
    }
    g.inv[((size_t)n * LD * LD + e) * 2] = sr;
    g.inv[((size_t)n * LD * LD + e) * 2 + 1] = si;
  }
  if (tid == 0) {
    double s = 0.0;
    for (int m = 0; m < D; ++m) s += log(lam[m]);
    g.logdet[n] = s;  // cacg.py:151
  }
}

// ------------------------------------------------------------------ weighted covariance
struct GenCov {
  const void* y;
  int layout;
  int64_t B;
  int T, D, K;
  const double* gamma;     // (B,K,T) (mask_b_stride elements between problems) or null (ones, K = 1)
  int64_t gamma_bstride;
  const double* q;         // (B,K,T) or null (ones)
  const double* saliency;  // (B,T) or null
  int mode;                // 0 M-step (cacg.py:310-327), 1 PSD normalised mask, 2 PSD plain sums,
                           // 3 Watson M-step (unit-norm frames, mask gamma sal, / sum of the mask)
  int weight_mode;
  double* out_cov;         // c128 (B,K,D,D)
  double* out_weight;      // (B,K) or null
  double* out_sum;         // (B,K) class sums or null
  int32_t* out_zero;       // (B) or null: 1 when the bin holds an all-zero frame
  int k0, kc;              // this launch accumulates classes k0 .. k0 + kc - 1 (kc <= KM)
};

// Work split of gen_cov: the upper triangle of C (C is Hermitian) is cut into register tiles
// of kTi x kTj entries; a thread owns one tile for a subset of the frames ("frame group"), so
// one frame costs kTi + kTj LDS reads for kTi * kTj entries, and the groups are summed in a
// fixed order at the end.
constexpr int kTi = 4, kTj = 2, kTe = kTi * kTj;

__device__ __forceinline__ int cov_tile_count(int D) {
  const int nbi = (D + kTi - 1) / kTi, nbj = (D + kTj - 1) / kTj;
  int n = 0;
  for (int bi = 0; bi < nbi; ++bi) n += max(0, nbj - bi * (kTi / kTj));
  return n;
}

// KM: compile-time bound of the classes accumulated per launch (accumulator registers), 3 or
// kCovMaxK; more classes are covered by further launches (k0)
template <int DP, int KM, typename YS>
__global__ void __launch_bounds__(kGenThreads) gen_cov_kernel(GenCov a) {
  constexpr int LDY = DP + 1;  // row stride of the frame tile in complex numbers (bank spread)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* ytile = reinterpret_cast<double*>(smem);            // [kTile][LDY][2], reused below
  double* wtile = ytile + gen_cov_stage_doubles(DP);          // [KM][kTile] weights of the chunk
  double* spart = wtile + (size_t)kCovMaxK * kTile;           // [K][kTile] class-sum partials
  double* csum = spart + (size_t)kGenMaxK * kTile;            // [K]
  __shared__ int zero_seen;
  const int tid = threadIdx.x;
  const int64_t b = blockIdx.x;
  const int D = a.D, K = a.K, T = a.T, k0 = a.k0, kc = a.kc;
  if (tid == 0) zero_seen = 0;
  for (int e = tid; e < K * kTile; e += kGenThreads) spart[e] = 0.0;
  // tile of this thread: tiles (bi, bj) with bj >= bi * kTi / kTj touch the upper triangle
  const int ntiles = cov_tile_count(D);
  const int G = min(kGenThreads / ntiles, kTile);  // frame groups
  const int tl = tid % ntiles, grp = tid / ntiles;
  const bool active = grp < G;
  int i0 = 0, j0 = 0;
  {
    const int nbj = (D + kTj - 1) / kTj;
    int r = tl;
    for (int bi = 0;; ++bi) {
      const int cnt = nbj - bi * (kTi / kTj);
      if (r < cnt) {
        i0 = bi * kTi;
        j0 = (bi * (kTi / kTj) + r) * kTj;
        break;
      }
      r -= cnt;
    }
  }
  double accr[kTe][KM], acci[kTe][KM];
#pragma unroll
  for (int e = 0; e < kTe; ++e)
#pragma unroll
    for (int k = 0; k < KM; ++k) {
      accr[e][k] = 0.0;
      acci[e][k] = 0.0;
    }
  for (int t0 = 0; t0 < T; t0 += kTile) {
    __syncthreads();
    // stage: thread (frame, channel) -> y; weights below
    for (int idx = tid; idx < kTile * DP; idx += kGenThreads) {
      const int tt = idx / DP, d = idx - tt * DP, t = t0 + tt;
      double re = 0.0, im = 0.0;
      if (t < T && d < D) load_y<YS>(a.y, a.layout, b, t, d, T, D, re, im);
      ytile[(tt * LDY + d) * 2] = re;
      ytile[(tt * LDY + d) * 2 + 1] = im;
    }
    __syncthreads();
    if (tid < kTile) {
      // frame norm (raw layout): the M-step sees unit-norm observations (cacgmm.py:216)
      const int t = t0 + tid;
      double n2 = 0.0;
      for (int d = 0; d < D; ++d) {
        const double re = ytile[(tid * LDY + d) * 2], im = ytile[(tid * LDY + d) * 2 + 1];
        n2 += re * re + im * im;
      }
      const double inv = (a.layout == PBBSS_LAYOUT_TD && (a.mode == 0 || a.mode == 3))
                             ? ((n2 > 0.0) ? 1.0 / n2 : 0.0) : 1.0;
      if (t < T && !(n2 > 0.0)) zero_seen = 1;  // benign race: every writer stores 1
      for (int k = 0; k < K; ++k) {
        double w = 0.0, gs = 0.0;
        if (t < T) {
          const double gm = a.gamma ? a.gamma[(size_t)b * a.gamma_bstride + (size_t)k * T + t] : 1.0;
          const double sal = a.saliency ? a.saliency[(size_t)b * T + t] : 1.0;
          gs = gm * sal;
          if (a.mode == 0) {
            const double qq = a.q ? a.q[((size_t)b * K + k) * T + t] : 1.0;
            w = gs / fmax(qq, 10.0 * kTiny) * inv;  // cacg.py:310, :322
          } else if (a.mode == 3) {
            w = gs * inv;  // complex_watson.py:279-281, :306-313: unit-norm frames
          } else {
            w = gs;
          }
        }
        if (k >= k0 && k < k0 + kc) wtile[(k - k0) * kTile + tid] = w;
        // class sums of ALL classes (the saliency form of the weights needs their total),
        // accumulated by the staging thread of the frame in its own LDS slot
        spart[k * kTile + tid] += gs;
      }
    }
    __syncthreads();
    if (active) {
      for (int tt = grp; tt < kTile; tt += G) {
        double ar[kTi], ai[kTi], br[kTj], bi[kTj];
#pragma unroll
        for (int x = 0; x < kTi; ++x) {
          ar[x] = ytile[(tt * LDY + i0 + x) * 2];
          ai[x] = ytile[(tt * LDY + i0 + x) * 2 + 1];
        }
#pragma unroll
        for (int x = 0; x < kTj; ++x) {
          br[x] = ytile[(tt * LDY + j0 + x) * 2];
          bi[x] = ytile[(tt * LDY + j0 + x) * 2 + 1];
        }
        double w[KM];
#pragma unroll
        for (int k = 0; k < KM; ++k) w[k] = (k < kc) ? wtile[k * kTile + tt] : 0.0;
#pragma unroll
        for (int e = 0; e < kTe; ++e) {
          const int x = e / kTj, z = e % kTj;
          const double pr = ar[x] * br[z] + ai[x] * bi[z];   // y_i conj(y_j)
          const double pi = ai[x] * br[z] - ar[x] * bi[z];
#pragma unroll
          for (int k = 0; k < KM; ++k) {
            if (k < kc) {
              accr[e][k] = fma(w[k], pr, accr[e][k]);
              acci[e][k] = fma(w[k], pi, acci[e][k]);
            }
          }
        }
      }
    }
  }
  // class sums: fixed-order sum of the kTile staging slots
  __syncthreads();
  if (tid < K) {
    double tot = 0.0;
    for (int x = 0; x < kTile; ++x) tot += spart[tid * kTile + x];
    csum[tid] = tot;
  }
  __syncthreads();
  double tot_abs = 0.0;
  for (int k = 0; k < K; ++k) tot_abs += fabs(csum[k]);
  // frame groups -> one sum per entry, class by class through the (now free) frame tile
  double* part = ytile;  // [kGenThreads][kTe][2]
#pragma unroll
  for (int kk = 0; kk < KM; ++kk) {
    if (kk < kc) {
      const int k = k0 + kk;
      __syncthreads();
      if (active) {
#pragma unroll
        for (int e = 0; e < kTe; ++e) {
          part[(tid * kTe + e) * 2] = accr[e][kk];
          part[(tid * kTe + e) * 2 + 1] = acci[e][kk];
        }
      }
      __syncthreads();
      double sc;
      if (a.mode == 0) sc = (double)D / fmax(csum[k], kTiny);        // cacg.py:316, :327
      else if (a.mode == 1) sc = 1.0 / fmax(csum[k], 1e-10);         // beamformer.py:123
      else if (a.mode == 3) sc = 1.0 / fmax(csum[k], kTiny);         // complex_watson.py:313
      else sc = a.gamma ? 1.0 : 1.0 / (double)T;                     // :114-117
      for (int o = tid; o < ntiles * kTe; o += kGenThreads) {
        const int tile = o / kTe, e = o - tile * kTe;
        // (i, j) of entry e of that tile: recompute the tile origin like above
        int ti0 = 0, tj0 = 0;
        {
          const int nbj = (D + kTj - 1) / kTj;
          int r = tile;
          for (int bi = 0;; ++bi) {
            const int cnt = nbj - bi * (kTi / kTj);
            if (r < cnt) {
              ti0 = bi * kTi;
              tj0 = (bi * (kTi / kTj) + r) * kTj;
              break;
            }
            r -= cnt;
          }
        }
        const int i = ti0 + e / kTj, j = tj0 + e % kTj;
        if (i < D && j < D && i <= j) {
          double sr = 0.0, si = 0.0;
          for (int gg = 0; gg < G; ++gg) {
            sr += part[((gg * ntiles + tile) * kTe + e) * 2];
            si += part[((gg * ntiles + tile) * kTe + e) * 2 + 1];
          }
          sr *= sc;
          si = (i == j) ? 0.0 : si * sc;
          double* up = a.out_cov + ((((size_t)b * K + k) * D + i) * D + j) * 2;
          double* lo = a.out_cov + ((((size_t)b * K + k) * D + j) * D + i) * 2;
          up[0] = sr;
          up[1] = si;
          lo[0] = sr;
          lo[1] = -si;
        }
      }
    }
  }
  if (tid == 0 && a.out_zero) a.out_zero[b] = zero_seen;
  if (tid >= k0 && tid < k0 + kc) {
    if (a.out_sum) a.out_sum[b * K + tid] = csum[tid];
    if (a.out_weight) {
      double w;
      if (a.weight_mode == PBBSS_WEIGHT_UNIFORM) w = 1.0 / K;
      else if (a.saliency) w = csum[tid] / ((tot_abs == 0.0) ? 1e-10 : tot_abs);  // mm_utils.py:192
      else w = csum[tid] / (double)T;                                               // :188
      a.out_weight[b * K + tid] = w;
    }
  }
}

// ------------------------------------------------------------------ M-step covariances (EM loop)
// C_k = D / (sum_t g_kt) * sum_t w_kt y_t y_t^H with the weights w of the E-step above
// (cacg.py:310-327), on raw (B, T, D) observations.  Matrix-free on the vector ALU with DPP
// operands: a wavefront owns an 8 x 16 block of C (rows i = 8 I + m, columns j = 16 J + c) and
// takes FOUR frames per instruction -- lane (r, c) = (frame of the quad, column):
//     own  : w_k(t_r) conj(y_j(t_r))   in the lane's registers
//     bcast: y_i(t_r) for the eight rows = 16 scalars of frame t_r in ONE register (lane c of
//            the row of 16 lanes holds scalar 16 I + c of the frame), read by the FMAs through
//            row_newbcast (pbbss_dev.hpp: fmac_row_bcast)
// i.e. per quad one complex load, one scalar load and K weights per lane, then 32 K FMAs; no
// LDS, no barrier, no cross-lane traffic inside the loop.  The four frame rows of a lane
// column are summed at the end, the four waves of the workgroup (frame quarters) through LDS
// in a fixed order.  Units (I, J) cover the upper triangle: I <= 2 J + 1.
struct GenCov2 {
  const void* y;
  int64_t B;
  int T, D, K;
  const double* mweight;  // (B,K,T)
  const double* csum;     // (B,K) class sums sum_t gamma sal
  double* out_cov;        // c128 (B,K,D,D)
  int k0, kc;             // classes of this launch (kc <= KM)
  int NI;                 // row slabs: ceil(D / 8)
};

template <int KM, typename YS>
__global__ void __launch_bounds__(kGenThreads) gen_cov2_kernel(GenCov2 a) {
  using YS2 = typename std::conditional<std::is_same<YS, float>::value, float2, double2>::type;
  __shared__ double part[kGenWaves][KM][8][16][2];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63, r = lane >> 4, c = lane & 15;
  const int64_t b = blockIdx.x;
  const int D = a.D, K = a.K, T = a.T, kc = a.kc;
  int I = blockIdx.y, J = 0;
  for (;; ++J) {
    const int cnt = min(2 * J + 2, a.NI);
    if (I < cnt) break;
    I -= cnt;
  }
  const int j = 16 * J + c, sidx = 16 * I + c;
  const bool jv = j < D, sv = sidx < 2 * D;
  const YS2* ybase = static_cast<const YS2*>(a.y) + (size_t)b * T * D;
  const double* wbase = a.mweight + ((size_t)b * K + a.k0) * T;
  const int nquad = (T + 3) / 4, qw = (nquad + kGenWaves - 1) / kGenWaves;
  const int q0 = wave * qw, q1 = min(q0 + qw, nquad);
  // raw loads at clamped addresses, masks when the values are used (a load inside its guard
  // compiles to a branch and a full wait per load)
  struct In {
    YS2 own;
    YS bc;
    double w[KM];
    bool tv;
  };
  const int jc = jv ? j : 0, sc = sv ? sidx : 0;
  auto load = [&](int q) {
    const int t = 4 * q + r;
    In x;
    x.tv = q < q1 && t < T;  // an invalid frame contributes through w = 0 only
    const int tc = x.tv ? t : 0;
    const YS2* fr = ybase + (size_t)tc * D;
    x.own = fr[jc];
    x.bc = reinterpret_cast<const YS*>(fr)[sc];
#pragma unroll
    for (int k = 0; k < KM; ++k) x.w[k] = wbase[(size_t)(k < kc ? k : 0) * T + tc];
    return x;
  };
  double accr[8][KM], acci[8][KM];
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int k = 0; k < KM; ++k) accr[m][k] = acci[m][k] = 0.0;
  auto accumulate = [&](const In& cur) {
    const double zr = jv ? (double)cur.own.x : 0.0, zi = jv ? (double)cur.own.y : 0.0;
    double breg = sv ? (double)cur.bc : 0.0;
    double u[KM], v[KM], nv[KM];
#pragma unroll
    for (int k = 0; k < KM; ++k) {
      const double wk = (cur.tv && k < kc) ? cur.w[k] : 0.0;
      u[k] = wk * zr;
      v[k] = wk * zi;
      nv[k] = -v[k];
    }
    // breg was just written by the vector ALU: two wait states before a DPP read of it
    asm volatile("s_nop 1" : "+v"(breg));
    static_for<0, 8>([&](auto mc) {
      constexpr int m = mc;
#pragma unroll
      for (int k = 0; k < KM; ++k) {
        // C_ij += w y_i conj(y_j): Re = re_i re_j + im_i im_j, Im = im_i re_j - re_i im_j
        fmac_row_bcast<2 * m>(accr[m][k], breg, u[k]);
        fmac_row_bcast<2 * m + 1>(accr[m][k], breg, v[k]);
        fmac_row_bcast<2 * m + 1>(acci[m][k], breg, u[k]);
        fmac_row_bcast<2 * m>(acci[m][k], breg, nv[k]);
      }
    });
  };
  // two quads per trip with ping-pong operand sets: a rolled `cur = nxt` makes the compiler
  // wait for the loads it has just issued before it can copy the registers
  In qa = load(q0), qb;
  for (int q = q0; q < q1; q += 2) {
    qb = load(q + 1);
    __builtin_amdgcn_sched_barrier(0);
    accumulate(qa);
    qa = load(q + 2);
    __builtin_amdgcn_sched_barrier(0);
    if (q + 1 < q1) accumulate(qb);  // uniform over the wave
  }
  // four frame rows of the wave, then the four waves
#pragma unroll
  for (int m = 0; m < 8; ++m) {
#pragma unroll
    for (int k = 0; k < KM; ++k) {
      double xr = accr[m][k], xi = acci[m][k];
      xr += __shfl_xor(xr, 16);
      xi += __shfl_xor(xi, 16);
      xr += __shfl_xor(xr, 32);
      xi += __shfl_xor(xi, 32);
      if (r == 0) {
        part[wave][k][m][c][0] = xr;
        part[wave][k][m][c][1] = xi;
      }
    }
  }
  __syncthreads();
  for (int e = tid; e < kc * 128; e += kGenThreads) {
    const int k = e >> 7, m = (e >> 4) & 7, cc = e & 15;
    const int i = 8 * I + m, jj = 16 * J + cc;
    if (i < D && jj < D && i <= jj) {
      double sr = 0.0, si = 0.0;
#pragma unroll
      for (int w = 0; w < kGenWaves; ++w) {
        sr += part[w][k][m][cc][0];
        si += part[w][k][m][cc][1];
      }
      const double sc = (double)D / fmax(a.csum[b * K + a.k0 + k], kTiny);  // cacg.py:316, :327
      sr *= sc;
      si = (i == jj) ? 0.0 : si * sc;
      double* up = a.out_cov + ((((size_t)b * K + a.k0 + k) * D + i) * D + jj) * 2;
      double* lo = a.out_cov + ((((size_t)b * K + a.k0 + k) * D + jj) * D + i) * 2;
      up[0] = sr;
      up[1] = si;
      lo[0] = sr;
      lo[1] = -si;
    }
  }
}

// class sums sum_t gamma_kt sal_t and the mixture weights (mixture_model_utils.py:180-201)
__global__ void __launch_bounds__(kGenThreads) gen_csum_kernel(const double* gamma,
                                                               const double* saliency, int K,
                                                               int T, int weight_mode,
                                                               double* out_sum,
                                                               double* out_weight) {
  __shared__ double red[kGenWaves];
  __shared__ double cs[kGenMaxK];
  const int tid = threadIdx.x;
  const int64_t b = blockIdx.x;
  for (int k = 0; k < K; ++k) {
    double acc = 0.0;
    for (int t = tid; t < T; t += kGenThreads)
      acc += gamma[((size_t)b * K + k) * T + t] * (saliency ? saliency[(size_t)b * T + t] : 1.0);
    acc = block_sum(acc, red, tid);
    if (tid == 0) cs[k] = acc;
    __syncthreads();
  }
  if (tid < K) {
    double tot_abs = 0.0;
    for (int k = 0; k < K; ++k) tot_abs += fabs(cs[k]);
    out_sum[b * K + tid] = cs[tid];
    if (out_weight) {
      double w;
      if (weight_mode == PBBSS_WEIGHT_UNIFORM) w = 1.0 / K;
      else if (saliency) w = cs[tid] / ((tot_abs == 0.0) ? 1e-10 : tot_abs);  // mm_utils.py:192
      else w = cs[tid] / (double)T;                                             // :188
      out_weight[b * K + tid] = w;
    }
  }
}

// M-step weights of an affiliation initialisation (first iteration: quadratic form = 1,
// cacgmm.py:211-228): w = gamma0 sal / |y|^2, and the all-zero-frame flag
template <typename YS>
__global__ void __launch_bounds__(kGenThreads) gen_init_weight_kernel(
    const void* y, int layout, int T, int D, int K, const double* gamma0, const double* saliency,
    double* out_mweight, int32_t* out_zero) {
  const int64_t b = blockIdx.x;
  const int t = blockIdx.y * kGenThreads + threadIdx.x;
  if (t >= T) return;
  // element strides of (frame, channel): (B, T, D) or the frame-contiguous (B, D, T) copy
  const size_t st = (layout == PBBSS_LAYOUT_TD) ? (size_t)D : 1;
  const size_t sd = (layout == PBBSS_LAYOUT_TD) ? 1 : (size_t)T;
  const YS* fr = static_cast<const YS*>(y) + 2 * ((size_t)b * T * D + (size_t)t * st);
  double n2 = 0.0;
  for (int d = 0; d < D; ++d) {
    const double re = (double)fr[2 * d * sd], im = (double)fr[2 * d * sd + 1];
    n2 += re * re + im * im;
  }
  const double inv = (n2 > 0.0) ? 1.0 / n2 : 0.0;
  if (!(n2 > 0.0) && out_zero) out_zero[b] = 1;
  const double sal = saliency ? saliency[(size_t)b * T + t] : 1.0;
  for (int k = 0; k < K; ++k) {
    const size_t idx = ((size_t)b * K + k) * T + t;
    out_mweight[idx] = gamma0[idx] * sal * inv;
  }
}

// ------------------------------------------------------------------ Hermitian eigensolver
struct GenHeev {
  const double* a;      // c128 (N,D,D)
  int64_t N;
  int D;
  int covariance_norm;  // -1: plain eigh; else PBBSS_COVNORM_* with the floor below
  double eig_floor;
  double* out_val;      // (N,D) ascending
  double* out_vec;      // c128 (N,D,D), eigenvectors in columns
  int32_t* out_status;  // (N) or null
  const int32_t* skip;  // (N) or null: nonzero = leave this matrix alone
};


// NT threads per matrix: one entry of the D x D block per thread at D = 32 (NT = 1024) -- the
// solver is a chain of ~200 rounds of three barriers each, and LDS (64 KB per matrix at DP = 32)
// admits only two workgroups per CU, so the waves that hide its latency must come from within
// the workgroup.
template <int DP, int NT>
__global__ void __launch_bounds__(NT) gen_heev_kernel(GenHeev g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* A = reinterpret_cast<double*>(smem);     // [DP][DP][2]
  double* A2 = A + DP * DP * 2;
  double* V = A2 + DP * DP * 2;
  double* V2 = V + DP * DP * 2;
  double* rot = V2 + DP * DP * 2;                  // [DP][3]: c, (s u) re, im of the pair of index x
  double* red = rot + DP * 3;                      // [NT / 64]
  int* part = reinterpret_cast<int*>(red + NT / kWave);  // [DP] partner of x in this round
  const int tid = threadIdx.x;
  const int64_t n = blockIdx.x;
  if (g.skip && g.skip[n]) return;  // uniform over the workgroup
  const int D = g.D;
  const int N = D + (D & 1);  // tournament size (even)
  int st = 0;
  // load (Hermitian from the upper triangle, like numpy's default UPLO='L' on the C-order
  // array == LAPACK upper of the transpose; both halves agree for the covariances used here)
  double tr = 0.0;
  for (int e = tid; e < DP * DP; e += NT) {
    const int i = e / DP, j = e - i * DP;
    double re = 0.0, im = 0.0;
    if (i < D && j < D) {
      const int lo = i < j ? j : i, hi = i < j ? i : j;  // read the LOWER triangle entry (lo,hi)
      const double* p = g.a + (((size_t)n * D + lo) * D + hi) * 2;
      re = p[0];
      im = (i == j) ? 0.0 : ((i > j) ? p[1] : -p[1]);
      if (i == j) tr += re;
    }
    A[e * 2] = re;
    A[e * 2 + 1] = im;
    V[e * 2] = (i == j) ? 1.0 : 0.0;
    V[e * 2 + 1] = 0.0;
  }
  tr = gen_block_sum<NT>(tr, red, tid);
  if (g.covariance_norm == PBBSS_COVNORM_TRACE) {  // cacg.py:88-90
    const double it = 1.0 / fmax(tr, kTiny);
    __syncthreads();
    for (int e = tid; e < DP * DP * 2; e += NT) A[e] *= it;
  }
  __syncthreads();
  double fro2 = 0.0;
  for (int e = tid; e < DP * DP; e += NT) fro2 += A[e * 2] * A[e * 2] + A[e * 2 + 1] * A[e * 2 + 1];
  fro2 = gen_block_sum<NT>(fro2, red, tid);
  if (!isfinite(fro2)) st |= PBBSS_ST_NONFINITE;
  const GenJacobiScratch scratch{A2, V2, rot, red, part};
  if (lds_jacobi_heev<NT>(A, V, scratch, D, DP, tid) < 0) st |= PBBSS_ST_EIG_NOCONV;
  __syncthreads();
  // eigenvalues -> rank (ascending, ties by index), normalisation and floor, outputs
  double* lam = A2;          // reuse: [DP] eigenvalues, [DP] processed
  int* rank = part;          // reuse
  if (tid < D) lam[tid] = A[(tid * DP + tid) * 2];
  __syncthreads();
  if (tid < D) {
    const double l = lam[tid];
    int rk = 0;
    double lmax = -1.79e308;
    for (int m = 0; m < D; ++m) {
      const double lm = lam[m];
      rk += (lm < l || (lm == l && m < tid)) ? 1 : 0;
      lmax = fmax(lmax, lm);
    }
    double lout = l;
    if (g.covariance_norm == PBBSS_COVNORM_EIGENVALUE) {  // cacg.py:112-121
      lout = l / fmax(lmax, kTiny);
      if (lout < g.eig_floor) {
        lout = g.eig_floor;
        st |= PBBSS_ST_FLOORED;
      }
    } else if (g.covariance_norm >= 0) {                   // cacg.py:122-126
      const double fl = lmax * g.eig_floor;
      if (lout < fl) {
        lout = fl;
        st |= PBBSS_ST_FLOORED;
      }
    }
    if (!isfinite(lout)) st |= PBBSS_ST_NONFINITE;
    rank[tid] = rk;
    g.out_val[(size_t)n * D + rk] = lout;
  }
  __syncthreads();
  for (int e = tid; e < DP * DP; e += NT) {
    const int i = e / DP, j = e - i * DP;
    if (i < D && j < D) {
      double* o = g.out_vec + (((size_t)n * D + i) * D + rank[j]) * 2;
      o[0] = V[e * 2];
      o[1] = V[e * 2 + 1];
    }
  }
  if (g.out_status) {
    // OR of the per-thread status words
    __shared__ int sst;
    if (tid == 0) sst = 0;
    __syncthreads();
    if (st) atomicOr(&sst, st);
    __syncthreads();
    if (tid == 0) g.out_status[n] = sst;
  }
}

// ------------------------------------------------------------------ Hermitian eigensolver, QL
// One WAVEFRONT per matrix: Householder reduction to a real symmetric tridiagonal matrix
// (LAPACK zhetd2's recurrences, reflectors H_k = I - tau_k v_k v_k^H kept below the
// subdiagonal), implicit-shift QL on (d, e) with the rotations accumulated in a REAL matrix Z
// (EISPACK tql2 / Numerical Recipes tqli), back-transformation V = H_0 ... H_{n-3} Z with one
// eigenvector per lane in registers, then the ordering / normalisation / floor of
// from_covariance (cacg.py:82-132).  ~5x fewer flops than the cyclic Jacobi of gen_heev_kernel
// and no workgroup barrier (a 64-thread workgroup's barrier is a wait, not a rendezvous); LDS
// per matrix: A n (n + 1) complex + Z n (n + 1) real + vectors = 21 KB at n = 29, seven
// matrices per CU.  Lane = column / row index; n <= DP <= 32 (DP sizes the register arrays).
template <int DP>
__global__ void __launch_bounds__(kWave) gen_heev_ql_kernel(GenHeev g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  const int64_t nm = blockIdx.x;
  if (g.skip && g.skip[nm]) return;  // uniform over the workgroup
  const int n = g.D, LD = n + 1;
  double* A = reinterpret_cast<double*>(smem);  // [n][LD][2]
  double* Zt = A + (size_t)n * LD * 2;          // [n][LD]   Zt[col][row]
  double* dv = Zt + (size_t)n * LD;             // [DP + 1] diagonal
  double* ev = dv + DP + 1;                     // [DP + 1] subdiagonal, e[k] couples k and k + 1
  double* tauv = ev + DP + 1;                   // [DP][2]
  double* vbuf = tauv + 2 * DP;                 // [DP][2] reflector of the current step
  double* wbuf = vbuf + 2 * DP;                 // [DP][2]
  int st = 0;
  // ---- load (lower triangle, like gen_heev_kernel), optional trace normalisation
  double tr = 0.0;
  for (int e = lane; e < n * n; e += kWave) {
    const int i = e / n, j = e - i * n;
    const int lo = i < j ? j : i, hi = i < j ? i : j;
    const double* p = g.a + (((size_t)nm * n + lo) * n + hi) * 2;
    const double re = p[0];
    const double im = (i == j) ? 0.0 : ((i > j) ? p[1] : -p[1]);
    if (i == j) tr += re;
    A[(i * LD + j) * 2] = re;
    A[(i * LD + j) * 2 + 1] = im;
    Zt[i * LD + j] = (i == j) ? 1.0 : 0.0;
  }
  tr = wave_sum(tr);
  __syncthreads();
  double fro2 = 0.0;
  {
    const double it = (g.covariance_norm == PBBSS_COVNORM_TRACE) ? 1.0 / fmax(tr, kTiny) : 1.0;  // cacg.py:88-90
    for (int e = lane; e < n * n; e += kWave) {
      const int i = e / n, j = e - i * n;
      const double re = A[(i * LD + j) * 2] * it, im = A[(i * LD + j) * 2 + 1] * it;
      A[(i * LD + j) * 2] = re;
      A[(i * LD + j) * 2 + 1] = im;
      fro2 += re * re + im * im;
    }
  }
  fro2 = wave_sum(fro2);
  if (!isfinite(fro2)) st |= PBBSS_ST_NONFINITE;
  __syncthreads();
  const bool solve = (fro2 > 0.0) && isfinite(fro2);
  double dreg, xre[DP], xim[DP];
  if (wave_heev_ql<DP>(A, LD, Zt, LD, dv, ev, tauv, vbuf, wbuf, n, lane, solve, dreg, xre, xim))
    st |= PBBSS_ST_EIG_NOCONV;
  // ---- eigenvalues -> rank (ascending, ties by index), normalisation and floor, outputs
  int rk = 0;
  double lmax = -1.79e308;
  for (int m = 0; m < n; ++m) {  // all lanes: the broadcast needs the whole wave
    const double lm = lane_bcast_const(dreg, m);
    rk += (lm < dreg || (lm == dreg && m < lane)) ? 1 : 0;
    lmax = fmax(lmax, lm);
  }
  if (lane < n) {
    const double l = dreg;
    double lout = l;
    if (g.covariance_norm == PBBSS_COVNORM_EIGENVALUE) {  // cacg.py:112-121
      lout = l / fmax(lmax, kTiny);
      if (lout < g.eig_floor) {
        lout = g.eig_floor;
        st |= PBBSS_ST_FLOORED;
      }
    } else if (g.covariance_norm >= 0) {                   // cacg.py:122-126
      const double fl = lmax * g.eig_floor;
      if (lout < fl) {
        lout = fl;
        st |= PBBSS_ST_FLOORED;
      }
    }
    if (!isfinite(lout)) st |= PBBSS_ST_NONFINITE;
    g.out_val[(size_t)nm * n + rk] = lout;
#pragma unroll
    for (int r = 0; r < DP; ++r) {
      if (r < n) {
        double* o = g.out_vec + (((size_t)nm * n + r) * n + rk) * 2;
        o[0] = xre[r];
        o[1] = xim[r];
      }
    }
  }
  if (g.out_status) {
    // OR over the wave
    int acc = st;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc |= __shfl_xor(acc, off);
    if (lane == 0) g.out_status[nm] = acc;
  }
}

// ------------------------------------------------------------------ inverse (EM fast path)
struct GenInv {
  const double* a;     // c128 (N,D,D)
  int64_t N;
  int D;
  double eig_floor;
  double* out_inv;     // c128 (N,D,D)
  double* out_logdet;  // (N)
  int32_t* out_ok;     // (N)
  const int32_t* veto; // (N / K) or null: nonzero = never accept (see launch_gen_inverse)
  int K;
  int LD;              // row stride of out_inv (gen_state_ld(D)), zero beyond D
};

// The matrix lives in REGISTERS for the whole sweep: thread (bi, bj) owns the BS x BS block of
// entries (bi BS + r, bj BS + c) (BS = 2 for DP > 16: 256 threads x 4 entries at DP = 32; BS = 3
// at DP = 36, D = 33 / 34: 144 threads x 9 entries); per
// pivot only the pivot row and column travel through LDS (double-buffered by pivot parity:
// one barrier per pivot), instead of the whole matrix being read and written there.
template <int DP>
__global__ void __launch_bounds__(kGenThreads) gen_inv_kernel(GenInv g) {
  constexpr int BS = (DP > 32) ? 3 : ((DP > 16) ? 2 : 1);  // DP = 36: 12 x 12 blocks of 3 x 3
  constexpr int NBK = DP / BS;  // blocks per dimension
  static_assert(NBK * NBK <= kGenThreads && DP % BS == 0, "one block per thread");
  __shared__ __attribute__((aligned(16))) double A[DP * DP * 2];  // only for the final symmetrisation
  __shared__ __attribute__((aligned(16))) double rowp[2][DP * 2];
  __shared__ __attribute__((aligned(16))) double colp[2][DP * 2];
  __shared__ double pivinv[2];  // reciprocal of the current pivot, by pivot parity
  __shared__ double pivs[DP];   // the pivots, for the log-determinant
  __shared__ double red[kGenWaves];
  const int tid = threadIdx.x;
  const int64_t n = blockIdx.x;
  const int D = g.D;
  const bool owner = tid < NBK * NBK;
  const int bi = owner ? tid / NBK : 0, bj = owner ? tid % NBK : 0;
  double ar[BS][BS], ai[BS][BS];
  double tr = 0.0;
#pragma unroll
  for (int r = 0; r < BS; ++r) {
#pragma unroll
    for (int c = 0; c < BS; ++c) {
      const int i = bi * BS + r, j = bj * BS + c;
      double re = 0.0, im = 0.0;
      if (owner && i < D && j < D) {  // same triangle as gen_heev
        const int lo = i < j ? j : i, hi = i < j ? i : j;
        const double* p = g.a + (((size_t)n * D + lo) * D + hi) * 2;
        re = p[0];
        im = (i == j) ? 0.0 : ((i > j) ? p[1] : -p[1]);
        if (i == j) tr += re;
      }
      ar[r][c] = re;
      ai[r][c] = im;
    }
  }
  tr = block_sum(tr, red, tid);  // ends with a barrier
  // Gauss-Jordan sweep without pivoting: the pivots of a Hermitian positive definite matrix
  // are its (positive) Schur complements and their product is the determinant
  bool ok = true;
  // pivot p = pb BS + pr: pr is a compile-time constant of the unrolled inner loop, so that
  // the owner's register arrays are indexed statically (a runtime row select is turned into an
  // indexed scratch access by hipcc)
  // Everything that is not the complex multiply-add of an entry is hoisted out of the sweep: the
  // reciprocal of the pivot is computed by the thread that owns it and travels with the pivot
  // row, the logarithms of the determinant are taken after the sweep (one pivot per thread), the
  // padding masks are loop invariants (57.5 -> 51.9 us at D = 29, 22.4 -> 15.7 us at D <= 16).
  // Measured and not adopted: 2 x 2 block pivots (half the publish / barrier / read round trips,
  // 60.7 us) and one wavefront per matrix with 4 x 4 register blocks (53.0 us, 25.7 us at
  // D <= 16); without the sweep the kernel takes 13.5 us.
  bool live[BS][BS];
#pragma unroll
  for (int r = 0; r < BS; ++r)
#pragma unroll
    for (int c = 0; c < BS; ++c) live[r][c] = (bi * BS + r < D) && (bj * BS + c < D);
  for (int pb = 0; pb < NBK; ++pb) {
    static_for<0, BS>([&](auto prc) {
      constexpr int pr = prc;
      const int p = pb * BS + pr;
      if (p >= D || !ok) return;  // uniform over the workgroup
      double* rp = rowp[p & 1];
      double* cp = colp[p & 1];
      if (owner && bi == pb) {
#pragma unroll
        for (int c = 0; c < BS; ++c) {
          rp[(bj * BS + c) * 2] = ar[pr][c];
          rp[(bj * BS + c) * 2 + 1] = ai[pr][c];
        }
        if (bj == pb) {  // owner of the pivot: its reciprocal (0 flags a bad pivot) and its value
          const double piv = ar[pr][pr];
          const bool good = (piv > 0.0) && isfinite(piv);
          pivinv[p & 1] = good ? 1.0 / piv : 0.0;
          pivs[p] = piv;
        }
      }
      if (owner && bj == pb) {
#pragma unroll
        for (int r = 0; r < BS; ++r) {
          cp[(bi * BS + r) * 2] = ar[r][pr];
          cp[(bi * BS + r) * 2 + 1] = ai[r][pr];
        }
      }
      __syncthreads();
      const double ip = pivinv[p & 1];
      if (ip == 0.0) {  // the same value in every thread
        ok = false;
        return;
      }
      // One straight-line update per entry, a' = a~ - (col~ . row~) / piv with a~ = 0 on row /
      // column p, col~_i = -1 on row p and row~_j = +1 on column p: it reproduces the four
      // textbook rules (pivot -> 1/piv, pivot row -> row/piv, pivot column -> -col/piv, rest
      // -> Schur update) without branches (wave_la.hpp uses the same form for D <= 8)
      double rr[BS], ri[BS];
#pragma unroll
      for (int c = 0; c < BS; ++c) {
        const int j = bj * BS + c;
        const bool jp = (j == p);
        rr[c] = (jp ? 1.0 : rp[j * 2]) * ip;
        ri[c] = (jp ? 0.0 : rp[j * 2 + 1]) * ip;
      }
#pragma unroll
      for (int r = 0; r < BS; ++r) {
        const int i = bi * BS + r;
        const bool irow = (i == p);
        const double cr = irow ? -1.0 : cp[i * 2], ci = irow ? 0.0 : cp[i * 2 + 1];
#pragma unroll
        for (int c = 0; c < BS; ++c) {
          const int j = bj * BS + c;
          const bool keep = !(irow || j == p);
          const double br = keep ? ar[r][c] : 0.0, bim = keep ? ai[r][c] : 0.0;
          const double xr = br - (cr * rr[c] - ci * ri[c]);
          const double xi = bim - (cr * ri[c] + ci * rr[c]);
          ar[r][c] = live[r][c] ? xr : 0.0;
          ai[r][c] = live[r][c] ? xi : 0.0;
        }
      }
    });
  }
  // log det = sum of the logarithms of the pivots, in pivot order
  double logdet = 0.0;
  __syncthreads();
  if (ok && tid < D) pivs[tid] = log(pivs[tid]);
  __syncthreads();
  if (ok && tid == 0) {
    for (int q = 0; q < D; ++q) logdet += pivs[q];
  }
  double fro2 = 0.0;
  if (ok && owner) {
#pragma unroll
    for (int r = 0; r < BS; ++r)
#pragma unroll
      for (int c = 0; c < BS; ++c) fro2 += ar[r][c] * ar[r][c] + ai[r][c] * ai[r][c];
  }
  if (owner) {
#pragma unroll
    for (int r = 0; r < BS; ++r)
#pragma unroll
      for (int c = 0; c < BS; ++c) {
        const int e = (bi * BS + r) * DP + bj * BS + c;
        A[e * 2] = ar[r][c];
        A[e * 2 + 1] = ai[r][c];
      }
  }
  fro2 = block_sum(fro2, red, tid);  // barrier: A complete
  // lambda_min >= 1 / ||C^-1||_F and lambda_max <= tr C (cacgmm_em.hpp uses the same test)
  const double bound = tr * sqrt(fro2);
  ok = ok && isfinite(bound) && (bound * g.eig_floor < 1e-2) && (bound < 1e13);
  if (g.veto && g.veto[n / g.K]) ok = false;
  if (ok) {
    const int LD = g.LD;
    for (int e = tid; e < LD * LD; e += kGenThreads) {
      const int i = e / LD, j = e - i * LD;
      double re = 0.0, im = 0.0;
      if (i < D && j < D) {
        // average the two triangles: the sweep keeps them conjugate only up to rounding
        re = 0.5 * (A[(i * DP + j) * 2] + A[(j * DP + i) * 2]);
        im = (i == j) ? 0.0 : 0.5 * (A[(i * DP + j) * 2 + 1] - A[(j * DP + i) * 2 + 1]);
      }
      double* o = g.out_inv + ((size_t)n * LD * LD + e) * 2;
      o[0] = re;
      o[1] = im;
    }
  }
  if (tid == 0) {
    g.out_logdet[n] = logdet;
    g.out_ok[n] = ok ? 1 : 0;
  }
}

inline int ok_or_hip() { return hipGetLastError() == hipSuccess ? PBBSS_OK : PBBSS_ERR_HIP; }

template <typename KFN>
int set_lds(KFN kfn, size_t lds, size_t lds_limit) {
  if (lds > lds_limit) return PBBSS_ERR_LDS_CAPACITY;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return PBBSS_ERR_HIP;
  return PBBSS_OK;
}

}  // namespace

bool gen_supported(int D, int K) { return D >= 2 && D <= kGenMaxD && K >= 1 && K <= kGenMaxK; }
bool gen_em_supported(int D, int K) { return D >= 2 && D <= kGenMaxDEm && K >= 1 && K <= kGenMaxK; }

int launch_gen_estep(const void* y, int y_is_c128, int layout, int64_t B, int T, int D, int K,
                     const double* eigvec, const double* eigval, const double* weight, int64_t wb,
                     int64_t wk, int64_t wt, const uint8_t* activity, double eps, double* out_aff,
                     double* out_q, double* out_logpdf, hipStream_t s,
                     const GenInverseState& state, const double* saliency, double* out_mweight,
                     int32_t* out_zero, int raw_dt, const double* extra, double spatial_scale) {
  if (!gen_em_supported(D, K)) return PBBSS_ERR_UNSUPPORTED;
  if (!state.inv || !state.logdet) return PBBSS_ERR_INVALID_ARG;
  const int DP = gen_state_ld(D);
  GenEigToInv c{eigvec, eigval, D, DP, state.ok, state.inv, state.logdet};
  hipLaunchKernelGGL(gen_eig_to_inv_kernel, dim3((unsigned)(B * K)), dim3(kGenThreads), 0, s, c);
  GenEstep a{y, layout, B, T, D, K, state.inv, state.logdet, weight, wb, wk, wt, activity, eps,
             out_aff, out_q, out_logpdf, saliency, out_mweight, out_zero,
             (layout == PBBSS_LAYOUT_TD || raw_dt) ? 1 : 0, extra, spatial_scale};
  const dim3 grid((unsigned)B, (unsigned)((T + kGenThreads - 1) / kGenThreads));
  if (grid.y > 65535u) return PBBSS_ERR_UNSUPPORTED;
  const size_t lds = (size_t)2 * K * kGenThreads * sizeof(double);  // softmax slots [2][K][thread]
#define PBBSS_GEN_E(DPV, YST)                                                              \
  {                                                                                        \
    auto kfn = gen_estep_kernel<DPV, YST>;                                                 \
    if (lds > 32768 && set_lds(kfn, lds, 131072) != PBBSS_OK) return PBBSS_ERR_HIP;        \
    hipLaunchKernelGGL(kfn, grid, dim3(kGenThreads), lds, s, a);                           \
  }
#define PBBSS_GEN_ED(DPV) \
  case DPV: if (y_is_c128) { PBBSS_GEN_E(DPV, double) } else { PBBSS_GEN_E(DPV, float) } break;
  switch (DP) {
    PBBSS_GEN_ED(12) PBBSS_GEN_ED(16) PBBSS_GEN_ED(20) PBBSS_GEN_ED(24) PBBSS_GEN_ED(28)
    PBBSS_GEN_ED(32) PBBSS_GEN_ED(36)
    default: return PBBSS_ERR_UNSUPPORTED;
  }
#undef PBBSS_GEN_ED
#undef PBBSS_GEN_E
  return ok_or_hip();
}

int launch_gen_cov(const void* y, int y_is_c128, int layout, int64_t B, int T, int D, int K,
                   const double* gamma, int64_t gamma_bstride, const double* q,
                   const double* saliency, int mode, int weight_mode, double* out_cov,
                   double* out_weight, double* out_sum, size_t lds_limit, hipStream_t s,
                   int32_t* out_zero) {
  if (!gen_em_supported(D, K)) return PBBSS_ERR_UNSUPPORTED;
  const int DP = D <= 16 ? 16 : (D <= 32 ? 32 : 36);
  const size_t lds = (gen_cov_stage_doubles(DP) + (size_t)kCovMaxK * kTile +
                      (size_t)kGenMaxK * kTile + kGenMaxK) * sizeof(double);
  int rc;
#define PBBSS_GEN_C(DPV, KMV, YST)                                                         \
  {                                                                                        \
    auto kfn = gen_cov_kernel<DPV, KMV, YST>;                                              \
    if ((rc = set_lds(kfn, lds, lds_limit)) != PBBSS_OK) return rc;                        \
    hipLaunchKernelGGL(kfn, dim3((unsigned)B), dim3(kGenThreads), lds, s, a);              \
  }
#define PBBSS_GEN_CK(DPV, YST) \
  if (kc <= 3) PBBSS_GEN_C(DPV, 3, YST) else PBBSS_GEN_C(DPV, kCovMaxK, YST)
  // balanced chunks of at most kCovMaxK classes, e.g. K = 9 -> 5 + 4, K = 7 -> 4 + 3
  const int nchunk = (K + kCovMaxK - 1) / kCovMaxK;
  for (int c = 0, k0 = 0; c < nchunk; ++c) {
    const int kc = (K - k0 + (nchunk - c) - 1) / (nchunk - c);
    GenCov a{y, layout, B, T, D, K, gamma, gamma_bstride, q, saliency, mode, weight_mode,
             out_cov, out_weight, out_sum, out_zero, k0, kc};
    if (DP == 16) {
      if (y_is_c128) { PBBSS_GEN_CK(16, double) } else { PBBSS_GEN_CK(16, float) }
    } else if (DP == 32) {
      if (y_is_c128) { PBBSS_GEN_CK(32, double) } else { PBBSS_GEN_CK(32, float) }
    } else {
      if (y_is_c128) { PBBSS_GEN_CK(36, double) } else { PBBSS_GEN_CK(36, float) }
    }
    k0 += kc;
  }
#undef PBBSS_GEN_CK
#undef PBBSS_GEN_C
  return ok_or_hip();
}

namespace {
// Inline permutation alignment of the joint models at generic sizes
// (mixture_model_utils.py:58-130; the fused kernel's phase_joint_pa does the same for D <= 8):
// per bin the class permutation of the (weighted) spatial log-pdf that maximises
// sum_{k,t} softmax_k(lp)(t) lp_k(t), lp = spatial[perm] + spectral (no mixture weights),
// itertools.permutations order, the first strict maximum wins; then the posterior with that
// permutation (log_pdf_to_affiliation with the weights) and the cACG M-step weights
// gamma sal / max(q, 10 tiny) / |y|^2 -- q stays in the spatial model's class order, as the
// reference hands predict's quadratic form to the M-step unpermuted (gcacgmm.py:98-117).
constexpr int kPaMaxK = 6;    // 720 permutations
constexpr int kPaBatch = 8;   // permutations scored per block reduction
struct GenJointPa {
  const void* yt;            // (B, D, T) raw observation (frame-contiguous copy)
  int T, D, K;
  const double* lp_spatial;  // (B,K,T) cACG log-pdf (gen_estep out_logpdf)
  const double* q;           // (B,K,T) quadratic forms
  const double* extra;       // (B,K,T) spectral log-pdf, already times spectral_weight
  double spatial_scale;
  const double* weight;
  int64_t wb, wk, wt;
  const double* saliency;    // (B,T) or null
  double eps;
  double* out_aff;           // (B,K,T)
  double* out_mweight;       // (B,K,T) or null
  int32_t* out_zero;         // (B) or null
  const uint8_t* activity;   // (B,K,T) source_activity_mask or null (stand-alone entry point)
  int32_t* out_perm;         // (B,K) chosen permutation or null
};

__device__ __forceinline__ void pa_nth_permutation(int p, int K, int* perm) {
  int fact = 1;
  for (int i = 2; i < K; ++i) fact *= i;  // (K-1)!
  unsigned used = 0;
  for (int i = 0; i < K; ++i) {
    const int d = p / fact;
    p -= d * fact;
    if (i < K - 1) fact /= (K - 1 - i);
    int pick = 0;
    for (int c = 0, seen = 0; c < K; ++c) {
      if (!(used >> c & 1)) {
        if (seen == d) pick = c;
        ++seen;
      }
    }
    used |= 1u << pick;
    perm[i] = pick;
  }
}

template <typename YS>
__global__ void __launch_bounds__(kGenThreads) gen_joint_pa_kernel(GenJointPa a) {
  __shared__ double red[kGenWaves][kPaBatch];
  __shared__ int best_perm[kPaMaxK];
  __shared__ int zero_seen;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t b = blockIdx.x;
  const int K = a.K, T = a.T, D = a.D;
  const double* sp = a.lp_spatial + (size_t)b * K * T;
  const double* ex = a.extra + (size_t)b * K * T;
  if (tid == 0) zero_seen = 0;
  int nperm = 1;
  for (int i = 2; i <= K; ++i) nperm *= i;
  double best = -INFINITY;
  int best_p = 0;
  for (int p0 = 0; p0 < nperm; p0 += kPaBatch) {
    double part[kPaBatch];
#pragma unroll
    for (int x = 0; x < kPaBatch; ++x) part[x] = 0.0;
    for (int x = 0; x < kPaBatch && p0 + x < nperm; ++x) {
      int perm[kPaMaxK];
      pa_nth_permutation(p0 + x, K, perm);
      double acc = 0.0;
      for (int t = tid; t < T; t += kGenThreads) {
        double lp[kPaMaxK], mx = -1.79e308;
        for (int k = 0; k < K; ++k) {
          lp[k] = fma(a.spatial_scale, sp[(size_t)perm[k] * T + t], ex[(size_t)k * T + t]);
          mx = fmax(mx, lp[k]);
        }
        double den = 0.0, num = 0.0;
        for (int k = 0; k < K; ++k) {
          const double e = exp(lp[k] - mx);
          den += e;
          num = fma(e, lp[k], num);
        }
        acc += num / fmax(den, kTiny);
      }
      part[x] = acc;
    }
    __syncthreads();  // the previous batch's reads of red are done
#pragma unroll
    for (int x = 0; x < kPaBatch; ++x) {
      const double v = wave_sum(part[x]);
      if (lane == 0) red[wave][x] = v;
    }
    __syncthreads();
    for (int x = 0; x < kPaBatch && p0 + x < nperm; ++x) {
      double tot = 0.0;
      for (int w = 0; w < kGenWaves; ++w) tot += red[w][x];
      if (tot > best) {  // strict: the first maximiser wins, as in the reference loop
        best = tot;
        best_p = p0 + x;
      }
    }
  }
  if (tid == 0) pa_nth_permutation(best_p, K, best_perm);
  __syncthreads();
  if (a.out_perm && tid < K) a.out_perm[b * K + tid] = best_perm[tid];
  // yt == null: the stand-alone posterior (pbbss_log_pdf_to_affiliation_inline_pa), no M-step weights
  const YS* y = a.yt ? static_cast<const YS*>(a.yt) + (size_t)b * D * T * 2 : nullptr;
  for (int t = tid; t < T; t += kGenThreads) {
    double inv = 0.0;
    if (y) {
      double n2 = 0.0;
      for (int d = 0; d < D; ++d) {
        const double re = (double)y[((size_t)d * T + t) * 2], im = (double)y[((size_t)d * T + t) * 2 + 1];
        n2 += re * re + im * im;
      }
      inv = (n2 > 0.0) ? 1.0 / n2 : 0.0;
      if (!(n2 > 0.0)) zero_seen = 1;  // benign race: every writer stores 1
    }
    double lp[kPaMaxK], mx = -1.79e308;
    for (int k = 0; k < K; ++k) {
      lp[k] = fma(a.spatial_scale, sp[(size_t)best_perm[k] * T + t], ex[(size_t)k * T + t]);
      mx = fmax(mx, lp[k]);
    }
    double v[kPaMaxK], den = 0.0;
    for (int k = 0; k < K; ++k) {
      v[k] = exp(lp[k] - mx) * a.weight[b * a.wb + k * a.wk + (int64_t)t * a.wt];
      if (a.activity) v[k] *= (double)a.activity[((size_t)b * K + k) * T + t];  // mmu.py:39-41
      den += v[k];
    }
    den = fmax(den, kTiny);
    const double sal = a.saliency ? a.saliency[(size_t)b * T + t] : 1.0;
    for (int k = 0; k < K; ++k) {
      double gam = v[k] / den;
      if (a.eps != 0.0) gam = fmin(fmax(gam, a.eps), 1.0 - a.eps);
      a.out_aff[((size_t)b * K + k) * T + t] = gam;
      if (a.out_mweight)
        a.out_mweight[((size_t)b * K + k) * T + t] =
            gam * sal / fmax(a.q[((size_t)b * K + k) * T + t], 10.0 * kTiny) * inv;
    }
  }
  __syncthreads();
  if (tid == 0 && a.out_zero && zero_seen) a.out_zero[b] = 1;
}
}  // namespace

int launch_gen_joint_pa(const void* yt, int y_is_c128, int64_t B, int T, int D, int K,
                        const double* lp_spatial, const double* q, const double* extra,
                        double spatial_scale, const double* weight, int64_t wb, int64_t wk,
                        int64_t wt, const double* saliency, double eps, double* out_aff,
                        double* out_mweight, int32_t* out_zero, hipStream_t s,
                        const uint8_t* activity, int32_t* out_perm) {
  if (K < 1 || K > kPaMaxK || B > 2147483647LL) return PBBSS_ERR_UNSUPPORTED;
  GenJointPa a{yt, T, D, K, lp_spatial, q, extra, spatial_scale, weight, wb, wk, wt, saliency, eps,
               out_aff, out_mweight, out_zero, activity, out_perm};
  if (y_is_c128)
    hipLaunchKernelGGL(gen_joint_pa_kernel<double>, dim3((unsigned)B), dim3(kGenThreads), 0, s, a);
  else
    hipLaunchKernelGGL(gen_joint_pa_kernel<float>, dim3((unsigned)B), dim3(kGenThreads), 0, s, a);
  return ok_or_hip();
}

namespace {
// (B, T, D) -> (B, D, T), values untouched: the E-step of the EM loop reads the observation with
// lane = frame, which on the raw layout is a stride of D complex numbers per lane (every load
// instruction touches 64 cache lines and the 15 KB footprint of a wave does not survive in L1:
// 18 500 of a wave's 63 000 cycles at D = 29 were this load phase); transposed once per fit, the
// frame axis is contiguous and every iteration's loads are coalesced.
template <typename YS2>
__global__ void __launch_bounds__(kGenThreads) gen_transpose_kernel(const YS2* y, int T, int D,
                                                                  YS2* out) {
  __shared__ YS2 tile[64][33];
  const int64_t b = blockIdx.y;
  const int t0 = blockIdx.x * 64;
  const YS2* src = y + (size_t)b * T * D;
  YS2* dst = out + (size_t)b * D * T;
  for (int d0 = 0; d0 < D; d0 += 32) {
    const int nd = min(32, D - d0);
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * nd; i += kGenThreads) {
      const int f = i / nd, c = i - f * nd;
      if (t0 + f < T) tile[f][c] = src[(size_t)(t0 + f) * D + d0 + c];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * nd; i += kGenThreads) {
      const int c = i >> 6, f = i & 63;
      if (t0 + f < T) dst[(size_t)(d0 + c) * T + t0 + f] = tile[f][c];
    }
  }
}
}  // namespace

int launch_gen_transpose(const void* y, int y_is_c128, int64_t B, int T, int D, void* out,
                         hipStream_t s) {
  if (B > 65535) return PBBSS_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)((T + 63) / 64), (unsigned)B);
  if (y_is_c128)
    hipLaunchKernelGGL(gen_transpose_kernel<double2>, grid, dim3(kGenThreads), 0, s,
                       static_cast<const double2*>(y), T, D, static_cast<double2*>(out));
  else
    hipLaunchKernelGGL(gen_transpose_kernel<float2>, grid, dim3(kGenThreads), 0, s,
                       static_cast<const float2*>(y), T, D, static_cast<float2*>(out));
  return ok_or_hip();
}

int launch_gen_init_weights(const void* y, int y_is_c128, int layout, int64_t B, int T, int D, int K,
                            const double* gamma0, const double* saliency, double* out_mweight,
                            int32_t* out_zero, hipStream_t s) {
  if (!gen_em_supported(D, K)) return PBBSS_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)B, (unsigned)((T + kGenThreads - 1) / kGenThreads));
  if (grid.y > 65535u) return PBBSS_ERR_UNSUPPORTED;
  if (y_is_c128)
    hipLaunchKernelGGL(gen_init_weight_kernel<double>, grid, dim3(kGenThreads), 0, s, y, layout, T, D,
                       K, gamma0, saliency, out_mweight, out_zero);
  else
    hipLaunchKernelGGL(gen_init_weight_kernel<float>, grid, dim3(kGenThreads), 0, s, y, layout, T, D,
                       K, gamma0, saliency, out_mweight, out_zero);
  return ok_or_hip();
}

int launch_gen_mstep_cov(const void* y, int y_is_c128, int64_t B, int T, int D, int K,
                         const double* mweight, const double* gamma, const double* saliency,
                         int weight_mode, double* csum, double* out_cov, double* out_weight,
                         hipStream_t s) {
  if (!gen_em_supported(D, K)) return PBBSS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(gen_csum_kernel, dim3((unsigned)B), dim3(kGenThreads), 0, s, gamma, saliency,
                     K, T, weight_mode, csum, out_weight);
  const int NI = (D + 7) / 8, NJ = (D + 15) / 16;
  int units = 0;
  for (int J = 0; J < NJ; ++J) units += (2 * J + 2 < NI) ? 2 * J + 2 : NI;
  const dim3 grid((unsigned)B, (unsigned)units);
  // three classes per launch (48 accumulator registers per lane: three waves per SIMD; six
  // classes at once need 256 VGPRs and leave one wave per SIMD); more classes = more launches
  constexpr int kChunk = 3;
  for (int k0 = 0; k0 < K; k0 += kChunk) {
    const int kc = (K - k0 < kChunk) ? K - k0 : kChunk;
    GenCov2 a{y, B, T, D, K, mweight, csum, out_cov, k0, kc, NI};
    if (y_is_c128) hipLaunchKernelGGL((gen_cov2_kernel<kChunk, double>), grid, dim3(kGenThreads), 0, s, a);
    else hipLaunchKernelGGL((gen_cov2_kernel<kChunk, float>), grid, dim3(kGenThreads), 0, s, a);
  }
  return ok_or_hip();
}

int launch_gen_heev(const double* a, int64_t N, int D, int covariance_norm, double eig_floor,
                    double* out_val, double* out_vec, int32_t* out_status, size_t lds_limit,
                    hipStream_t s, const int32_t* skip) {
  if (D < 2 || D > kGenMaxDEm) return PBBSS_ERR_UNSUPPORTED;
  GenHeev g{a, N, D, covariance_norm, eig_floor, out_val, out_vec, out_status, skip};
  const int DP = D <= 16 ? 16 : (D <= 24 ? 24 : (D <= 32 ? 32 : 36));
  int rc;
  static const bool use_jacobi = [] {
    const char* v = getenv("PBBSS_GEN_HEEV");
    return v && v[0] == 'j';
  }();
  if (!use_jacobi || D > kGenMaxD) {  // tridiagonal QL, one wavefront per matrix (the only solver beyond 32)
    const size_t lds = ((size_t)D * (D + 1) * 3 + 2 * (DP + 1) + 6 * DP) * sizeof(double);
#define PBBSS_GEN_Q(DPV)                                                          \
  {                                                                               \
    auto kfn = gen_heev_ql_kernel<DPV>;                                           \
    if ((rc = set_lds(kfn, lds, lds_limit)) != PBBSS_OK) return rc;               \
    hipLaunchKernelGGL(kfn, dim3((unsigned)N), dim3(kWave), lds, s, g);           \
  }
    if (DP == 16) PBBSS_GEN_Q(16)
    else if (DP == 24) PBBSS_GEN_Q(24)
    else if (DP == 32) PBBSS_GEN_Q(32)
    else PBBSS_GEN_Q(36)
#undef PBBSS_GEN_Q
    return ok_or_hip();
  }
#define PBBSS_GEN_H(DPV, NTV)                                                                  \
  {                                                                                            \
    const size_t lds = ((size_t)4 * DPV * DPV * 2 + DPV * 3 + NTV / 64) * sizeof(double) +    \
                       DPV * sizeof(int);                                                      \
    auto kfn = gen_heev_kernel<DPV, NTV>;                                                      \
    if ((rc = set_lds(kfn, lds, lds_limit)) != PBBSS_OK) return rc;                            \
    hipLaunchKernelGGL(kfn, dim3((unsigned)N), dim3(NTV), lds, s, g);                          \
  }
  if (DP == 16) PBBSS_GEN_H(16, 256)
  else if (DP == 24) PBBSS_GEN_H(24, 576)
  else PBBSS_GEN_H(32, 1024)
#undef PBBSS_GEN_H
  return ok_or_hip();
}

int launch_gen_inverse(const double* a, int64_t N, int D, double eig_floor, double* out_inv,
                       double* out_logdet, int32_t* out_ok, hipStream_t s, const int32_t* veto,
                       int K) {
  if (D < 2 || D > kGenMaxDEm || K < 1) return PBBSS_ERR_UNSUPPORTED;
  GenInv g{a, N, D, eig_floor, out_inv, out_logdet, out_ok, veto, K, gen_state_ld(D)};
  if (D <= 16) {
    hipLaunchKernelGGL(gen_inv_kernel<16>, dim3((unsigned)N), dim3(kGenThreads), 0, s, g);
  } else if (D <= 24) {
    hipLaunchKernelGGL(gen_inv_kernel<24>, dim3((unsigned)N), dim3(kGenThreads), 0, s, g);
  } else if (D <= 32) {
    hipLaunchKernelGGL(gen_inv_kernel<32>, dim3((unsigned)N), dim3(kGenThreads), 0, s, g);
  } else {
    hipLaunchKernelGGL(gen_inv_kernel<36>, dim3((unsigned)N), dim3(kGenThreads), 0, s, g);
  }
  return ok_or_hip();
}

}  // namespace pbbss
