// Remaining members of the reference's beamformer family:
// LCMV, phase correction along frequency, distortionless / zero-degree normalisation,
// SNR post-filter, covariance conditioning, time-varying ("online") filter application.
// Reference: extraction/beamformer.py:414-456, :491-599.
//
// All of these are O(F D^2) touch-once operations on a few hundred small matrices: one
// wavefront per matrix where a solve is involved (wave_la.hpp), one thread per
// frequency / frame otherwise; no staging, the whole problem sits in L2.
#include "beamform.hpp"
#include "pbbss_dev.hpp"
#include "wave_la.hpp"

namespace pbbss {
namespace {

constexpr int kLaThreads = 256;
constexpr int kLaWaves = kLaThreads / kWave;

struct Cx {
  double re, im;
};
__device__ __forceinline__ Cx cmul(Cx a, Cx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ Cx cmulc(Cx a, Cx b) {  // conj(a) * b
  return {a.re * b.re + a.im * b.im, a.re * b.im - a.im * b.re};
}
__device__ __forceinline__ Cx cdiv(Cx a, Cx b) {
  // Smith's algorithm, as C99 / NumPy divide complex numbers
  if (fabs(b.re) >= fabs(b.im)) {
    double r = b.im / b.re, d = b.re + b.im * r;
    return {(a.re + a.im * r) / d, (a.im - a.re * r) / d};
  }
  double r = b.re / b.im, d = b.re * r + b.im;
  return {(a.re * r + a.im) / d, (a.im * r - a.re) / d};
}
__device__ __forceinline__ Cx ld(const double* p, size_t i) { return {p[2 * i], p[2 * i + 1]}; }
__device__ __forceinline__ void st(double* p, size_t i, Cx v) {
  p[2 * i] = v.re;
  p[2 * i + 1] = v.im;
}

// ------------------------------------------------------------------ LCMV
// w_f = P t,  P = Phi_f^-1 H_f (D x K),  (H_f^H P) t = response     (beamformer.py:430-454)
// atf (K,F,D), response (K), noise (F,D,D), all c128.  One wavefront per frequency;
// lane (i,j) of the 8x8 grid.  The K x K system is embedded in D x D with a diagonal pad
// of the system's own scale, so the shared LU / pseudo-inverse routines apply unchanged.
template <int D>
__global__ void __launch_bounds__(kLaThreads)
    lcmv_kernel(const double* atf, const double* response, const double* noise, int64_t F, int K,
                double* out_w, int32_t* status) {
  const int lane = threadIdx.x & 63;
  const int64_t f = (int64_t)blockIdx.x * kLaWaves + (threadIdx.x >> 6);
  if (f >= F) return;
  const LaneIJ c = lane_ij(lane);
  double nre = 0.0, nim = 0.0, bre = 0.0, bim = 0.0;
  if (c.i < D && c.j < D) {
    const double* p = noise + ((f * D + c.i) * D + c.j) * 2;
    nre = p[0];
    nim = p[1];
  }
  if (c.i < D && c.j < K) {  // B = H: column k is the ATF of source k
    const double* p = atf + (((size_t)c.j * F + f) * D + c.i) * 2;
    bre = p[0];
    bim = p[1];
  }
  double pre, pim;
  bool sing = wave_lu_solve<D>(nre, nim, bre, bim, c, pre, pim);
  if (sing) wave_pinv_solve<D>(nre, nim, bre, bim, c, pre, pim);  // stable_solve, math/solve.py:111
  if (!(c.i < D && c.j < K)) {
    pre = 0.0;
    pim = 0.0;
  }
  // G = H^H P: left factor A_kd = conj(H_dk) on lane (k, d)
  double are = 0.0, aim = 0.0;
  if (c.i < K && c.j < D) {
    const double* p = atf + (((size_t)c.i * F + f) * D + c.j) * 2;
    are = p[0];
    aim = -p[1];
  }
  double gre, gim;
  wave_matmul<D>(are, aim, pre, pim, c, gre, gim);
  const double pad = lane_get(gre, ij_lane(0, 0));
  if (c.i >= K || c.j >= K) {
    gre = (c.i == c.j && c.i < D) ? ((pad != 0.0 && isfinite(pad)) ? fabs(pad) : 1.0) : 0.0;
    gim = 0.0;
  }
  double rre = 0.0, rim = 0.0;
  if (c.j == 0 && c.i < K) {
    // the reference casts the response to complex64 (beamformer.py:443)
    rre = (double)(float)response[2 * c.i];
    rim = (double)(float)response[2 * c.i + 1];
  }
  double tre, tim;
  bool sing2 = wave_lu_solve<D>(gre, gim, rre, rim, c, tre, tim);
  if (sing2) wave_pinv_solve<D>(gre, gim, rre, rim, c, tre, tim);
  if (!(c.j == 0 && c.i < K)) {
    tre = 0.0;
    tim = 0.0;
  }
  double wre, wim;
  wave_matmul<D>(pre, pim, tre, tim, c, wre, wim);
  if (c.j == 0 && c.i < D) {
    out_w[(f * D + c.i) * 2] = wre;
    out_w[(f * D + c.i) * 2 + 1] = wim;
  }
  if (status && lane == 0) status[f] = (sing || sing2) ? PBBSS_ST_SINGULAR : 0;
}

// ------------------------------------------------------------------ phase correction
// u[l, r] = exp(j angle(sum_d conj(v[.., f, d]) v[.., f-1, d])), f >= 1   (beamformer.py:551-559)
// v viewed as (L, R, F, D): L = leading axis (the reference's cumprod runs along axis 0 of
// the (.., F-1, 1) array: the frequency axis for 2-D input, the FIRST axis otherwise).
__global__ void phase_unit_kernel(const double* v, int64_t M, int F, int D, double* u) {
  // one thread per (m, f), m over all leading axes; u (M, F-1)
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * (F - 1)) return;
  const int64_t m = i / (F - 1);
  const int f = (int)(i - m * (F - 1)) + 1;
  Cx s{0.0, 0.0};
  for (int d = 0; d < D; ++d) {
    Cx a = ld(v, ((size_t)m * F + f) * D + d), b = ld(v, ((size_t)m * F + f - 1) * D + d);
    Cx p = cmulc(a, b);
    s.re += p.re;
    s.im += p.im;
  }
  const double ang = atan2(s.im, s.re);  // np.angle; angle(0) = 0
  u[2 * i] = cos(ang);
  u[2 * i + 1] = sin(ang);
}

// cumulative product along the scan axis and application; u viewed as (L, R) with L the
// scan length: thread r walks l = 0..L-1.
//   2-D input : L = F-1, R = 1, vector row = l + 1
//   N-D input : L = shape[0], R = prod(rest) * (F-1); element (l, r) scales row (l, r/(F-1), r%(F-1)+1)
__global__ void phase_scan_kernel(const double* v, const double* u, int64_t L, int64_t R, int F,
                                  int D, int two_d, double* out) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  Cx acc{1.0, 0.0};
  for (int64_t l = 0; l < L; ++l) {
    acc = cmul(acc, ld(u, (size_t)l * R + r));
    size_t row;
    if (two_d) {
      row = (size_t)l + 1;
    } else {
      const int64_t rest = r / (F - 1);
      const int f = (int)(r - rest * (F - 1)) + 1;
      row = ((size_t)l * (R / (F - 1)) + rest) * F + f;
    }
    for (int d = 0; d < D; ++d) st(out, row * D + d, cmul(ld(v, row * D + d), acc));
  }
}

// 2-D input (F, D): the scan runs along the frequency axis, R = 1 -- one thread of the kernel
// above would walk all F-1 bins, a chain of ~F dependent L2 round trips (0.5 ms at 513 bins).
// One workgroup instead: thread i multiplies its chunk of consecutive factors, the 256 chunk
// products are scanned through LDS (Hillis-Steele, complex products of unit-modulus numbers:
// the order of the products differs from np.cumprod by rounding only), then every thread
// replays its chunk from the exclusive prefix and scales its rows.
constexpr int kScanThreads = 256;
__global__ void __launch_bounds__(kScanThreads)
    phase_scan_2d_kernel(const double* v, const double* u, int64_t L, int D, double* out) {
  __shared__ Cx part[2][kScanThreads];
  const int i = threadIdx.x;
  const int64_t chunk = (L + kScanThreads - 1) / kScanThreads;
  const int64_t l0 = chunk * i, l1 = (l0 + chunk < L) ? l0 + chunk : L;
  Cx acc{1.0, 0.0};
  for (int64_t l = l0; l < l1; ++l) acc = cmul(acc, ld(u, (size_t)l));
  part[0][i] = acc;
  __syncthreads();
  int cur = 0;
  for (int off = 1; off < kScanThreads; off <<= 1) {
    Cx x = part[cur][i];
    if (i >= off) x = cmul(part[cur][i - off], x);
    part[cur ^ 1][i] = x;
    cur ^= 1;
    __syncthreads();
  }
  acc = (i == 0) ? Cx{1.0, 0.0} : part[cur][i - 1];
  for (int64_t l = l0; l < l1; ++l) {
    acc = cmul(acc, ld(u, (size_t)l));
    const size_t row = (size_t)l + 1;
    for (int d = 0; d < D; ++d) st(out, row * D + d, cmul(ld(v, row * D + d), acc));
  }
}

// ------------------------------------------------------------------ per-frequency scalars
// mode 0: mvdr_snr_postfilter  (w^H T w) / (w^H N w)                    (beamformer.py:502-509)
// mode 1: distortionless_normalization  (N w)(w^H a) / (w^H N w)        (:491-499)
// One lane per (bin, row a): DP = next power of two >= D lanes form a bin's group (64 / DP bins per
// wavefront); a lane reads ROW a of the matrices (contiguous, neighbouring lanes neighbouring rows)
// and the group sums over the rows with xor shuffles.  (Until round 5: one thread per bin walking
// D x D strided entries -- 513 threads for the whole call.)
__device__ __forceinline__ Cx group_sum(Cx v, int DP) {
  for (int off = DP >> 1; off > 0; off >>= 1) {
    v.re += __shfl_xor(v.re, off);
    v.im += __shfl_xor(v.im, off);
  }
  return v;
}
__global__ void bf_quadratic_kernel(int mode, const double* w, const double* m1, const double* m2,
                                    const double* atf, int64_t F, int D, int DP, double* out) {
  const int lane = threadIdx.x & 63;
  const int a = lane & (DP - 1);
  const int64_t f = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / DP;
  const bool on = f < F && a < D;
  const int64_t fc = f < F ? f : F - 1;  // idle lanes take part in the shuffles with zeros
  Cx r2{0.0, 0.0}, r1{0.0, 0.0}, wa{0.0, 0.0};
  if (on) {
    wa = ld(w, (size_t)fc * D + a);
    const double* M2 = m2 + ((size_t)fc * D + a) * D * 2;
    const double* M1 = m1 ? m1 + ((size_t)fc * D + a) * D * 2 : nullptr;
    for (int b = 0; b < D; ++b) {
      const Cx wb = ld(w, (size_t)fc * D + b);
      const Cx p = cmul(ld(M2, (size_t)b), wb);
      r2.re += p.re;
      r2.im += p.im;
      if (mode == 0) {
        const Cx q = cmul(ld(M1, (size_t)b), wb);
        r1.re += q.re;
        r1.im += q.im;
      }
    }
  }
  const Cx den = group_sum(cmulc(wa, r2), DP);  // w^H N w
  if (mode == 0) {
    const Cx num = group_sum(cmulc(wa, r1), DP);  // w^H T w
    if (on && a == 0) st(out, (size_t)f, cdiv(num, den));
    return;
  }
  const Cx proj = group_sum(on ? cmulc(wa, ld(atf, (size_t)fc * D + a)) : Cx{0.0, 0.0}, DP);  // w^H a
  if (on) st(out, (size_t)f * D + a, cmul(cdiv(r2, den), proj));
}

// zero_degree_normalization: v * exp(-j angle(v[..., ref]))              (beamformer.py:512-514)
// ------------------------------------------------------------------ reference channel, rank one
// get_optimal_reference_channel (beamformer.py:601-624): per bin and candidate channel r the
// two quadratic forms  w_r^H T w_r  and  w_r^H N w_r  of the r-th COLUMN of the filter matrix.
// One thread per (bin, r); w_mat (F,D,D), T, N (F,D,D) c128 -> num, den (F,D) c128.
__global__ void refch_terms_kernel(const double* wm, const double* tp, const double* nn, int64_t F,
                                   int D, double* num, double* den) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= F * D) return;
  const int64_t f = i / D;
  const int r = (int)(i - f * D);
  const double* W = wm + (size_t)f * D * D * 2;
  const double* T = tp + (size_t)f * D * D * 2;
  const double* N = nn + (size_t)f * D * D * 2;
  Cx sn{0.0, 0.0}, sd{0.0, 0.0};
  for (int a = 0; a < D; ++a) {
    const Cx wa = ld(W, (size_t)a * D + r);
    Cx rt{0.0, 0.0}, rn{0.0, 0.0};
    for (int b = 0; b < D; ++b) {
      const Cx wb = ld(W, (size_t)b * D + r);
      const Cx pt = cmul(ld(T, (size_t)a * D + b), wb), pn = cmul(ld(N, (size_t)a * D + b), wb);
      rt.re += pt.re;
      rt.im += pt.im;
      rn.re += pn.re;
      rn.im += pn.im;
    }
    const Cx ct = cmulc(wa, rt), cn = cmulc(wa, rn);
    sn.re += ct.re;
    sn.im += ct.im;
    sd.re += cn.re;
    sd.im += cn.im;
  }
  st(num, (size_t)i, sn);
  st(den, (size_t)i, sd);
}

// rank-one approximation scaled to the trace of the covariance (beamformer_wrapper.py:18-25,
// :61-69):  out = a a^H * tr(C) / tr(a a^H).  One thread per (problem, i, j).
__global__ void rank_one_kernel(const double* cov, const double* a, int64_t N, int D, double* out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * D * D) return;
  const int64_t n = i / (D * D);
  const int ij = (int)(i - n * D * D), r = ij / D, c = ij - r * D;
  Cx trc{0.0, 0.0};
  double tra = 0.0;
  for (int d = 0; d < D; ++d) {
    const Cx cd = ld(cov, ((size_t)n * D + d) * D + d), ad = ld(a, (size_t)n * D + d);
    trc.re += cd.re;
    trc.im += cd.im;
    tra += ad.re * ad.re + ad.im * ad.im;
  }
  const Cx scale = cdiv(trc, Cx{tra, 0.0});
  const Cx ar = ld(a, (size_t)n * D + r), ac = ld(a, (size_t)n * D + c);
  const Cx outer{ar.re * ac.re + ar.im * ac.im, ar.im * ac.re - ar.re * ac.im};  // a_r conj(a_c)
  st(out, (size_t)i, cmul(scale, outer));
}

// y = M x per problem (the ATF estimate Phi_nn w_gev, beamformer_wrapper.py:28-48).
__global__ void matvec_kernel(const double* m, const double* x, int64_t N, int D, double* y) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * D) return;
  const int64_t n = i / D;
  const int r = (int)(i - n * D);
  Cx acc{0.0, 0.0};
  for (int c = 0; c < D; ++c) {
    const Cx p = cmul(ld(m, ((size_t)n * D + r) * D + c), ld(x, (size_t)n * D + c));
    acc.re += p.re;
    acc.im += p.im;
  }
  st(y, (size_t)i, acc);
}

__global__ void zero_degree_kernel(const double* v, int64_t N, int D, int ref, double* out) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  Cx r = ld(v, (size_t)n * D + ref);
  const double ang = -atan2(r.im, r.re);
  const Cx rot{cos(ang), sin(ang)};
  for (int d = 0; d < D; ++d) st(out, (size_t)n * D + d, cmul(ld(v, (size_t)n * D + d), rot));
}

// condition_covariance: (x + gamma tr(x)/D I) / (1 + gamma)              (beamformer.py:563-569)
// one thread per matrix ENTRY (the D diagonal reads of the trace hit the cache lines the
// neighbouring threads stream)
__global__ void condition_kernel(const double* x, int64_t N, int D, double gamma, double* out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N * D * D) return;
  const int64_t n = e / (D * D);
  const int ab = (int)(e - n * D * D);
  const int a = ab / D, b = ab - a * D;
  Cx v = ld(x, (size_t)e);
  if (a == b) {
    Cx tr{0.0, 0.0};
    for (int d = 0; d < D; ++d) {
      const Cx t = ld(x, ((size_t)n * D + d) * D + d);
      tr.re += t.re;
      tr.im += t.im;
    }
    v.re += gamma * tr.re / D;
    v.im += gamma * tr.im / D;
  }
  const double inv = 1.0 + gamma;
  st(out, (size_t)e, Cx{v.re / inv, v.im / inv});
}

// apply_online_beamforming_vector: out[f,t] = sum_d conj(v[t,f,d]) mix[f,d,t]   (:586-598)
// Tile of 16 bins x 16 frames per workgroup, the bin index fastest: the frame-varying vectors
// v (T, F, D) are then read in runs of 16 D contiguous complex numbers, the observation (F, D, T)
// in runs of 16 frames.  (Until round 5: one bin per block row -- every lane of a wavefront read v
// with a stride of F D complex numbers.)
template <typename YS>
__global__ void apply_online_kernel(const double* v, const void* mixv, int64_t F, int T, int D,
                                    double* out) {
  const int fl = threadIdx.x & 15, tl = threadIdx.x >> 4;
  const int64_t f = (int64_t)blockIdx.y * 16 + fl;
  const int t = blockIdx.x * 16 + tl;
  if (t >= T || f >= F) return;
  const YS* mix = static_cast<const YS*>(mixv);
  Cx s{0.0, 0.0};
  for (int d = 0; d < D; ++d) {
    const size_t mi = ((size_t)f * D + d) * T + t;
    Cx m{(double)mix[2 * mi], (double)mix[2 * mi + 1]};
    Cx p = cmulc(ld(v, ((size_t)t * F + f) * D + d), m);
    s.re += p.re;
    s.im += p.im;
  }
  st(out, (size_t)f * T + t, s);
}

inline int ok_or_hip() { return hipGetLastError() == hipSuccess ? PBBSS_OK : PBBSS_ERR_HIP; }
inline unsigned blocks(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

}  // namespace

int launch_lcmv(const double* atf, const double* response, const double* noise, int64_t F, int D,
                int K, double* w, int32_t* st, hipStream_t s) {
  if (K < 1 || K > D) return PBBSS_ERR_INVALID_ARG;
#define PBBSS_LCMV_CASE(DD)                                                                     \
  case DD:                                                                                      \
    hipLaunchKernelGGL(lcmv_kernel<DD>, dim3(blocks(F, kLaWaves)), dim3(kLaThreads), 0, s, atf, \
                       response, noise, F, K, w, st);                                           \
    break;
  switch (D) {
    PBBSS_LCMV_CASE(2) PBBSS_LCMV_CASE(3) PBBSS_LCMV_CASE(4) PBBSS_LCMV_CASE(5)
    PBBSS_LCMV_CASE(6) PBBSS_LCMV_CASE(7) PBBSS_LCMV_CASE(8)
    default: return PBBSS_ERR_UNSUPPORTED;
  }
#undef PBBSS_LCMV_CASE
  return ok_or_hip();
}

int launch_phase_correction(const double* v, int64_t lead, int64_t rest, int F, int D, int two_d,
                            double* scratch_u, double* out, hipStream_t s) {
  // v (lead, rest, F, D); two_d: lead = rest = 1.  F == 1: plain copy.
  const int64_t M = lead * rest;
  if (hipMemcpyAsync(out, v, (size_t)M * F * D * 16, hipMemcpyDeviceToDevice, s) != hipSuccess)
    return PBBSS_ERR_HIP;
  if (F < 2) return PBBSS_OK;
  hipLaunchKernelGGL(phase_unit_kernel, dim3(blocks(M * (F - 1), 256)), dim3(256), 0, s, v, M, F,
                     D, scratch_u);
  const int64_t L = two_d ? (F - 1) : lead;
  const int64_t R = two_d ? 1 : rest * (F - 1);
  if (two_d) {
    hipLaunchKernelGGL(phase_scan_2d_kernel, dim3(1), dim3(kScanThreads), 0, s, v, scratch_u, L, D,
                       out);
  } else {
    hipLaunchKernelGGL(phase_scan_kernel, dim3(blocks(R, 64)), dim3(64), 0, s, v, scratch_u, L, R,
                       F, D, two_d, out);
  }
  return ok_or_hip();
}

int launch_bf_quadratic(int mode, const double* w, const double* m1, const double* m2,
                        const double* atf, int64_t F, int D, double* out, hipStream_t s) {
  if (D < 1 || D > 64) return PBBSS_ERR_UNSUPPORTED;
  int DP = 1;
  while (DP < D) DP <<= 1;  // lanes per bin
  hipLaunchKernelGGL(bf_quadratic_kernel, dim3(blocks(F * DP, 256)), dim3(256), 0, s, mode, w, m1,
                     m2, atf, F, D, DP, out);
  return ok_or_hip();
}

int launch_refch_terms(const double* wm, const double* tp, const double* nn, int64_t F, int D,
                       double* num, double* den, hipStream_t s) {
  hipLaunchKernelGGL(refch_terms_kernel, dim3(blocks(F * D, 256)), dim3(256), 0, s, wm, tp, nn, F, D,
                     num, den);
  return ok_or_hip();
}

int launch_rank_one(const double* cov, const double* a, int64_t N, int D, double* out, hipStream_t s) {
  hipLaunchKernelGGL(rank_one_kernel, dim3(blocks(N * D * D, 256)), dim3(256), 0, s, cov, a, N, D, out);
  return ok_or_hip();
}

int launch_matvec(const double* m, const double* x, int64_t N, int D, double* y, hipStream_t s) {
  hipLaunchKernelGGL(matvec_kernel, dim3(blocks(N * D, 256)), dim3(256), 0, s, m, x, N, D, y);
  return ok_or_hip();
}

int launch_zero_degree(const double* v, int64_t N, int D, int ref, double* out, hipStream_t s) {
  hipLaunchKernelGGL(zero_degree_kernel, dim3(blocks(N, 256)), dim3(256), 0, s, v, N, D, ref, out);
  return ok_or_hip();
}

int launch_condition_covariance(const double* x, int64_t N, int D, double gamma, double* out,
                                hipStream_t s) {
  hipLaunchKernelGGL(condition_kernel, dim3(blocks(N * D * D, 256)), dim3(256), 0, s, x, N, D, gamma,
                     out);
  return ok_or_hip();
}

int launch_apply_online(const double* v, const void* mix, int mix_is_c128, int64_t F, int T, int D,
                        double* out, hipStream_t s) {
  if (F > 65535 * 16) return PBBSS_ERR_UNSUPPORTED;
  dim3 grid(blocks(T, 16), blocks(F, 16));
  if (mix_is_c128)
    hipLaunchKernelGGL(apply_online_kernel<double>, grid, dim3(256), 0, s, v, mix, F, T, D, out);
  else
    hipLaunchKernelGGL(apply_online_kernel<float>, grid, dim3(256), 0, s, v, mix, F, T, D, out);
  return ok_or_hip();
}

}  // namespace pbbss
