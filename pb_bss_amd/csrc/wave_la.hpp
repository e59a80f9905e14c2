// Wave-cooperative dense linear algebra on ONE small complex matrix per
// wavefront (D <= 8): lane l owns entry (i, j) = (l >> 3, l & 7) of each matrix
// it works on, held in two float64 registers (re, im).  All exchange is
// cross-lane (ds_bpermute), no LDS memory and no barriers, so four waves of a
// workgroup can factor four matrices independently.
//
// Replaces the LAPACK calls on the reference hot path:
//   numpy.linalg.eigh      distribution/complex_angular_central_gaussian.py:95
//                          extraction/beamformer.py:180
//   zhegvd                 extraction/cythonized/get_gev_vector.pyx:124-129
//   numpy.linalg.solve     math/solve.py:96, extraction/beamformer.py:250
// Every routine must be called by all 64 lanes of the wave (convergent).
#pragma once
#include "pbbss_dev.hpp"

namespace pbbss {

struct LaneIJ {
  int i, j;
};
__device__ __forceinline__ LaneIJ lane_ij(int lane) { return {lane >> 3, lane & 7}; }
__device__ __forceinline__ int ij_lane(int i, int j) { return (i << 3) | j; }

// ---------------------------------------------------------------------------
// Cholesky  A = L L^H  (right-looking).  In: Hermitian A_ij.  Out: L_ij valid
// for i >= j (strict upper part is scratch).  det(A) = prod pivots is returned
// as mantissa/exponent.  Returns 0 on success, else 1 + index of the first
// non-positive (or non-finite) pivot -- LAPACK zpotrf INFO semantics.
// ---------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ int wave_cholesky(double& are, double& aim, LaneIJ c,
                                             ScaledReal& det) {
  int info = 0;
  det.m = 1.0;
  det.e = 0;
#pragma unroll
  for (int p = 0; p < D; ++p) {
    double d = lane_get(are, ij_lane(p, p));
    bool good = (d > 0.0) && (d < 1.79e308);
    if (!good && info == 0) info = p + 1;
    double ds = good ? d : 1.0;
    scaled_mul(det, ds);
    double rs = 1.0 / sqrt(ds);
    if (c.j == p && c.i >= p) {
      are *= rs;
      aim = (c.i == p) ? 0.0 : aim * rs;
    }
    double lir = lane_get(are, ij_lane(c.i, p)), lii = lane_get(aim, ij_lane(c.i, p));
    double ljr = lane_get(are, ij_lane(c.j, p)), lji = lane_get(aim, ij_lane(c.j, p));
    if (c.i > p && c.j > p) {  // a_ij -= L_ip * conj(L_jp)
      are -= lir * ljr + lii * lji;
      aim -= lii * ljr - lir * lji;
    }
  }
  return info;
}

// ---------------------------------------------------------------------------
// In-place Gauss-Jordan inverse of a Hermitian positive definite matrix (no
// pivoting needed: the pivots are the Cholesky pivots d_p > 0).  One sweep per
// column p:  a_pp <- 1/d, row p <- row/d, col p <- -col/d,
//            a_ij <- a_ij - a_ip a_pj / d.
// Per sweep: one SGPR broadcast of the pivot and ONE parallel cross-lane hop
// (row element a_pj and column element a_ip) -- about a third of the dependent
// hops of Cholesky + triangular inverse + Gram product.  det(A) = prod d_p is
// returned as mantissa/exponent; info = 0, or 1 if some pivot was bad.
// ---------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ int wave_hpd_inverse(double& are, double& aim, LaneIJ c,
                                                ScaledReal& det) {
  // One fused update per column p:  a' = a~ - (col~ * row~) / d  with
  //   a~_ij  = 0 on row p and column p, a_ij elsewhere
  //   col~_i = -1 on row p,    a_ip elsewhere
  //   row~_j = +1 on column p, a_pj elsewhere
  // which reproduces the four rules above (a_pp <- 1/d, row/d, -col/d, a_ij - a_ip a_pj/d) as
  // straight-line code: the selects and the complex product wait only for the cross-lane round
  // trip, so after the Newton-refined reciprocal of the pivot ONE fma per component remains on
  // the serial path.  A bad pivot (non-positive / non-finite) only raises a flag -- the sweep
  // runs on with garbage that the caller discards (exact eigen path).  Returns 0 or 1.
  bool bad = false;
  det.m = 1.0;
  det.e = 0;
#pragma unroll
  for (int p = 0; p < D; ++p) {
    const double rr = lane_get(are, ij_lane(p, c.j)), ri = lane_get(aim, ij_lane(p, c.j));  // a_pj
    const double cr = lane_get(are, ij_lane(c.i, p)), ci = lane_get(aim, ij_lane(c.i, p));  // a_ip
    const double d = lane_bcast_const(are, ij_lane(p, p));
    bad |= !((d > 0.0) && (d < 1.79e308));
    int ex;
    det.m *= frexp(d, &ex);  // mantissas in [0.5, 1): D <= 8 factors cannot underflow
    det.e += ex;
    const double inv = fast_rcp(d);
    const bool ip = (c.i == p), jp = (c.j == p);
    const double br = (ip || jp) ? 0.0 : are, bi = (ip || jp) ? 0.0 : aim;
    const double xr = ip ? -1.0 : cr, xi = ip ? 0.0 : ci;
    const double yr = jp ? 1.0 : rr, yi = jp ? 0.0 : ri;
    const double tr = xr * yr - xi * yi, ti = xr * yi + xi * yr;
    are = fma(-tr, inv, br);
    aim = fma(-ti, inv, bi);
  }
  int e2;
  det.m = frexp(det.m, &e2);
  det.e += e2;
  return bad ? 1 : 0;
}

// X = L^-1 for lower-triangular L (row-oriented forward substitution on I).
template <int D>
__device__ __forceinline__ void wave_tri_inverse(double lre, double lim, LaneIJ c,
                                                 double& xre, double& xim) {
  xre = (c.i == c.j) ? 1.0 : 0.0;
  xim = 0.0;
#pragma unroll
  for (int m = 0; m < D; ++m) {
    double rd = 1.0 / lane_get(lre, ij_lane(m, m));
    if (c.i == m) {
      xre *= rd;
      xim *= rd;
    }
    double xmr = lane_get(xre, ij_lane(m, c.j)), xmi = lane_get(xim, ij_lane(m, c.j));
    double lr = lane_get(lre, ij_lane(c.i, m)), li = lane_get(lim, ij_lane(c.i, m));
    if (c.i > m) {
      xre -= lr * xmr - li * xmi;
      xim -= lr * xmi + li * xmr;
    }
  }
}

// G = X^H X  (G_ij = sum_m conj(X_mi) X_mj)
template <int D>
__device__ __forceinline__ void wave_gram(double xre, double xim, LaneIJ c,
                                          double& gre, double& gim) {
  gre = 0.0;
  gim = 0.0;
#pragma unroll
  for (int m = 0; m < D; ++m) {
    double ar = lane_get(xre, ij_lane(m, c.i)), ai = lane_get(xim, ij_lane(m, c.i));
    double br = lane_get(xre, ij_lane(m, c.j)), bi = lane_get(xim, ij_lane(m, c.j));
    gre += ar * br + ai * bi;
    gim += ar * bi - ai * br;
  }
}

// C = A * B (all D x D), C_ij = sum_m A_im B_mj
template <int D>
__device__ __forceinline__ void wave_matmul(double are, double aim, double bre,
                                            double bim, LaneIJ c, double& cre,
                                            double& cim) {
  cre = 0.0;
  cim = 0.0;
#pragma unroll
  for (int m = 0; m < D; ++m) {
    double ar = lane_get(are, ij_lane(c.i, m)), ai = lane_get(aim, ij_lane(c.i, m));
    double br = lane_get(bre, ij_lane(m, c.j)), bi = lane_get(bim, ij_lane(m, c.j));
    cre += ar * br - ai * bi;
    cim += ar * bi + ai * br;
  }
}

// conjugate transpose: out_ij = conj(a_ji)
__device__ __forceinline__ void wave_adjoint(double are, double aim, LaneIJ c,
                                             double& ore, double& oim) {
  ore = lane_get(are, ij_lane(c.j, c.i));
  oim = -lane_get(aim, ij_lane(c.j, c.i));
}

// ---------------------------------------------------------------------------
// Hermitian eigendecomposition by parallel cyclic Jacobi: a round-robin
// tournament schedule zeroes D/2 disjoint off-diagonal pairs per round with
// unitary plane rotations  A <- J^H A J,  V <- V J.
// In: Hermitian A_ij.  Out: A diagonal (eigenvalues on lanes (j,j), unsorted),
// V_ij = i-th component of the eigenvector belonging to a_jj.
// Returns the number of sweeps used, or -1 if not converged in kMaxSweeps.
// ---------------------------------------------------------------------------
constexpr int kJacobiMaxSweeps = 24;
constexpr double kJacobiTol = 1e-29;      // off-diagonal Frobenius^2 / total Frobenius^2
constexpr double kJacobiTolLoose = 1e-24; // accepted if the sweep budget runs out

template <int D>
__device__ __forceinline__ int wave_jacobi_heev(double& are, double& aim, LaneIJ c,
                                                double& vre, double& vim) {
  constexpr int N = D + (D & 1);  // tournament size (even)
  const bool valid = (c.i < D) && (c.j < D);
  if (!valid) {
    are = 0.0;
    aim = 0.0;
  }
  if (c.i == c.j) aim = 0.0;
  vre = (c.i == c.j) ? 1.0 : 0.0;
  vim = 0.0;
  // Frobenius norm for the convergence threshold
  const double fro2 = wave_sum(are * are + aim * aim);
  if (!(fro2 > 0.0)) return 0;  // zero (or NaN) matrix: nothing to rotate
  int sweeps = -1;
  for (int sweep = 0; sweep < kJacobiMaxSweeps; ++sweep) {
    double off2 = wave_sum((c.i != c.j) ? (are * are + aim * aim) : 0.0);
    if (off2 <= kJacobiTol * fro2) {
      sweeps = sweep;
      break;
    }
#pragma unroll 1
    for (int r = 0; r < N - 1; ++r) {
      // partner of index x in round r of the circle method
      auto partner = [&](int x) -> int {
        if (x >= N) return x;  // padding lanes of the 8x8 lane grid
        if (x == N - 1) return r;
        if (x == r) return N - 1;
        int y = 2 * r - x;
        y %= (N - 1);
        if (y < 0) y += N - 1;
        return y;
      };
      // --- rotation for the pair that contains this lane's ROW index
      const int pi_ = partner(c.i);
      const int p = min(c.i, pi_), q = max(c.i, pi_);
      const bool live = (q < D) && (p != q);
      const int ps = live ? p : 0, qs = live ? q : 0;
      double app = lane_get(are, ij_lane(ps, ps));
      double aqq = lane_get(are, ij_lane(qs, qs));
      double xr = lane_get(are, ij_lane(ps, qs));
      double xi = lane_get(aim, ij_lane(ps, qs));
      double g2 = xr * xr + xi * xi;
      double cs = 1.0, sur = 0.0, sui = 0.0;  // c, s*u (u = a_pq/|a_pq|)
      if (live && g2 > 0.0) {
        // Jacobi angle without sqrt/divide sequences: with d = a_qq - a_pp, g = |a_pq| and
        // h = sqrt(d^2 + 4 g^2) the classical tau / t / c / s formulas reduce to
        //   c^2 = (h + |d|) / (2 h),   s u = sign(d) a_pq / (h c)      (sign(0) = +1)
        // i.e. two Newton-refined reciprocal square roots on the dependent path of a round.
        const double d = aqq - app;
        const double h2 = fma(d, d, 4.0 * g2);
        if (h2 > 0.0 && h2 < 1.79e308) {
          const double rh = fast_rsqrt(h2);
          const double c2 = fma(0.5 * fabs(d), rh, 0.5);
          const double rc = fast_rsqrt(c2);  // c2 in [0.5, 1]
          cs = c2 * rc;
          const double ig = ((d < 0.0) ? -rh : rh) * rc;
          sur = xr * ig;
          sui = xi * ig;
        }
      }
      // parameters of the pair containing this lane's COLUMN index live on lane (j, j)
      double cc = lane_get(cs, ij_lane(c.j, c.j));
      double cur = lane_get(sur, ij_lane(c.j, c.j));
      double cui = lane_get(sui, ij_lane(c.j, c.j));
      const int pj = partner(c.j);
      const bool col_is_p = c.j < pj;
      // --- row transform  B = J^H A : row p' = c A_p - s u A_q ; row q' = s conj(u) A_p + c A_q
      {
        double orr = lane_get(are, ij_lane(pi_ < 8 ? pi_ : c.i, c.j));
        double oii = lane_get(aim, ij_lane(pi_ < 8 ? pi_ : c.i, c.j));
        double nr, ni;
        if (c.i < pi_) {  // this lane is in row p: a = c a - (s u) o
          nr = cs * are - (sur * orr - sui * oii);
          ni = cs * aim - (sur * oii + sui * orr);
        } else {          // row q: a = (s conj u) o + c a
          nr = cs * are + (sur * orr + sui * oii);
          ni = cs * aim + (sur * oii - sui * orr);
        }
        are = nr;
        aim = ni;
      }
      // --- column transform  A' = B J : col p' = c B_p - s conj(u) B_q ; col q' = s u B_p + c B_q
      {
        double orr = lane_get(are, ij_lane(c.i, pj < 8 ? pj : c.j));
        double oii = lane_get(aim, ij_lane(c.i, pj < 8 ? pj : c.j));
        double vor = lane_get(vre, ij_lane(c.i, pj < 8 ? pj : c.j));
        double voi = lane_get(vim, ij_lane(c.i, pj < 8 ? pj : c.j));
        double nr, ni, wr, wi;
        if (col_is_p) {  // col p: a = c a - (s conj u) o
          nr = cc * are - (cur * orr + cui * oii);
          ni = cc * aim - (cur * oii - cui * orr);
          wr = cc * vre - (cur * vor + cui * voi);
          wi = cc * vim - (cur * voi - cui * vor);
        } else {         // col q: a = (s u) o + c a
          nr = cc * are + (cur * orr - cui * oii);
          ni = cc * aim + (cur * oii + cui * orr);
          wr = cc * vre + (cur * vor - cui * voi);
          wi = cc * vim + (cur * voi + cui * vor);
        }
        are = nr;
        aim = ni;
        vre = wr;
        vim = wi;
      }
      if (c.i == c.j) aim = 0.0;
    }
  }
  if (sweeps < 0) {  // budget exhausted: accept if at the rounding-noise floor
    double off2 = wave_sum((c.i != c.j) ? (are * are + aim * aim) : 0.0);
    if (off2 <= kJacobiTolLoose * fro2) sweeps = kJacobiMaxSweeps;
  }
  return sweeps;
}

// ---------------------------------------------------------------------------
// The same Jacobi with the per-round lane bookkeeping taken from a table.
// A round of wave_jacobi_heev is bound by its instruction count, not by its cross-lane hops
// (a one-batch variant with MORE instructions was slower), and about a quarter of
// those instructions recompute what depends on (round, lane) alone: the tournament partners,
// the pair's (p, q), the source lanes of the thirteen exchanges and their byte addresses.
// jacobi_table_build writes them once per kernel -- per round and lane two dwords:
//   w0 = byte addresses (lane * 4, as ds_bpermute takes them) of (p,p) | (q,q) | (p,q) |
//        (partner row, j);   w1 = byte address of (i, partner column) | flags << 8
//        (1: pair live, 2: this lane's row is the pair's p, 4: its column is the pair's p)
// -- and wave_jacobi_heev_tab reads them back with one ds_read_b64 per round.  The arithmetic
// is that of wave_jacobi_heev, statement by statement: identical results.
// ---------------------------------------------------------------------------
template <int D>
constexpr int jacobi_table_dwords() {
  return (D + (D & 1) - 1) * kWave * 2;
}

template <int D>
__device__ __forceinline__ void jacobi_table_build(uint32_t* tab, int lane) {
  constexpr int N = D + (D & 1);
  const LaneIJ c = lane_ij(lane);
  for (int r = 0; r < N - 1; ++r) {
    auto partner = [&](int x) -> int {
      if (x >= N) return x;
      if (x == N - 1) return r;
      if (x == r) return N - 1;
      int y = 2 * r - x;
      y %= (N - 1);
      if (y < 0) y += N - 1;
      return y;
    };
    const int pi_ = partner(c.i), pj = partner(c.j);
    const int p = min(c.i, pi_), q = max(c.i, pi_);
    const bool live = (q < D) && (p != q);
    const int ps = live ? p : 0, qs = live ? q : 0;
    const uint32_t w0 = (uint32_t)(4 * ij_lane(ps, ps)) | (uint32_t)(4 * ij_lane(qs, qs)) << 8 |
                        (uint32_t)(4 * ij_lane(ps, qs)) << 16 |
                        (uint32_t)(4 * ij_lane(pi_ < 8 ? pi_ : c.i, c.j)) << 24;
    const uint32_t flags = (live ? 1u : 0u) | ((c.i < pi_) ? 2u : 0u) | ((c.j < pj) ? 4u : 0u);
    const uint32_t w1 = (uint32_t)(4 * ij_lane(c.i, pj < 8 ? pj : c.j)) | flags << 8;
    tab[(r * kWave + lane) * 2] = w0;
    tab[(r * kWave + lane) * 2 + 1] = w1;
  }
}

// value of lane (byte_addr / 4)
__device__ __forceinline__ double lane_get_addr(double v, int byte_addr) {
  F64Bits u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_ds_bpermute(byte_addr, u.i[0]);
  u.i[1] = __builtin_amdgcn_ds_bpermute(byte_addr, u.i[1]);
  return u.d;
}

template <int D>
__device__ __forceinline__ int wave_jacobi_heev_tab(double& are, double& aim, LaneIJ c,
                                                    double& vre, double& vim, const uint32_t* tab,
                                                    int lane) {
  constexpr int N = D + (D & 1);
  const bool valid = (c.i < D) && (c.j < D);
  if (!valid) {
    are = 0.0;
    aim = 0.0;
  }
  if (c.i == c.j) aim = 0.0;
  vre = (c.i == c.j) ? 1.0 : 0.0;
  vim = 0.0;
  const double fro2 = wave_sum(are * are + aim * aim);
  if (!(fro2 > 0.0)) return 0;
  const int jj = 4 * ij_lane(c.j, c.j);
  const uint2* tb = reinterpret_cast<const uint2*>(tab) + lane;
  int sweeps = -1;
  for (int sweep = 0; sweep < kJacobiMaxSweeps; ++sweep) {
    double off2 = wave_sum((c.i != c.j) ? (are * are + aim * aim) : 0.0);
    if (off2 <= kJacobiTol * fro2) {
      sweeps = sweep;
      break;
    }
#pragma unroll 1
    for (int r = 0; r < N - 1; ++r) {
      const uint2 e = tb[r * kWave];
      const int a_pp = e.x & 0xff, a_qq = (e.x >> 8) & 0xff, a_pq = (e.x >> 16) & 0xff;
      const int a_row = e.x >> 24, a_col = e.y & 0xff;
      const bool live = (e.y & 0x100) != 0, row_is_p = (e.y & 0x200) != 0;
      const bool col_is_p = (e.y & 0x400) != 0;
      double app = lane_get_addr(are, a_pp);
      double aqq = lane_get_addr(are, a_qq);
      double xr = lane_get_addr(are, a_pq);
      double xi = lane_get_addr(aim, a_pq);
      double g2 = xr * xr + xi * xi;
      double cs = 1.0, sur = 0.0, sui = 0.0;  // c, s*u (u = a_pq/|a_pq|)
      if (live && g2 > 0.0) {
        const double d = aqq - app;
        const double h2 = fma(d, d, 4.0 * g2);
        if (h2 > 0.0 && h2 < 1.79e308) {
          const double rh = fast_rsqrt(h2);
          const double c2 = fma(0.5 * fabs(d), rh, 0.5);
          const double rc = fast_rsqrt(c2);  // c2 in [0.5, 1]
          cs = c2 * rc;
          const double ig = ((d < 0.0) ? -rh : rh) * rc;
          sur = xr * ig;
          sui = xi * ig;
        }
      }
      double cc = lane_get_addr(cs, jj);
      double cur = lane_get_addr(sur, jj);
      double cui = lane_get_addr(sui, jj);
      {
        double orr = lane_get_addr(are, a_row);
        double oii = lane_get_addr(aim, a_row);
        double nr, ni;
        if (row_is_p) {
          nr = cs * are - (sur * orr - sui * oii);
          ni = cs * aim - (sur * oii + sui * orr);
        } else {
          nr = cs * are + (sur * orr + sui * oii);
          ni = cs * aim + (sur * oii - sui * orr);
        }
        are = nr;
        aim = ni;
      }
      {
        double orr = lane_get_addr(are, a_col);
        double oii = lane_get_addr(aim, a_col);
        double vor = lane_get_addr(vre, a_col);
        double voi = lane_get_addr(vim, a_col);
        double nr, ni, wr, wi;
        if (col_is_p) {
          nr = cc * are - (cur * orr + cui * oii);
          ni = cc * aim - (cur * oii - cui * orr);
          wr = cc * vre - (cur * vor + cui * voi);
          wi = cc * vim - (cur * voi - cui * vor);
        } else {
          nr = cc * are + (cur * orr - cui * oii);
          ni = cc * aim + (cur * oii + cui * orr);
          wr = cc * vre + (cur * vor - cui * voi);
          wi = cc * vim + (cur * voi + cui * vor);
        }
        are = nr;
        aim = ni;
        vre = wr;
        vim = wi;
      }
      if (c.i == c.j) aim = 0.0;
    }
  }
  if (sweeps < 0) {
    double off2 = wave_sum((c.i != c.j) ? (are * are + aim * aim) : 0.0);
    if (off2 <= kJacobiTolLoose * fro2) sweeps = kJacobiMaxSweeps;
  }
  return sweeps;
}

// ---------------------------------------------------------------------------
// Dominant eigenpair of a Hermitian positive semi-definite matrix from a good start vector
// (the complex-Watson M-step keeps ONE eigenpair, complex_watson.py:300-315 / utils.get_pca; a
// full Jacobi sweep set was 8.5-14.5 us of every EM iteration).
//
// Shifted inverse iteration that stays on positive definite systems: with the Rayleigh
// quotient rho of the current vector and its residual norm r there is an eigenvalue within r of
// rho; sigma = rho + 1.01 r + guard lies above it.  If that eigenvalue is the largest one,
// sigma I - C is positive definite, the pivot-free Gauss-Jordan inverse (wave_hpd_inverse) is
// stable, and two applications of the explicit inverse contract the error by
// ((sigma - l1) / (sigma - l2))^2.  Rounds repeat (new rho, new shift: cubic-like convergence)
// until the residual is at rounding level.  A non-positive pivot says that an eigenvalue LARGER
// than sigma exists (the start vector tracked the wrong eigenpair) -> return false, and so does
// a residual that does not reach the tolerance: the caller falls back to the full Jacobi
// decomposition.  A successful last inversion certifies lambda_max < sigma, i.e. the returned
// eigenvalue is the largest one up to the final residual.
//
// In:  A (lane (i,j) = entry), start vector x (xr, xi) with component i on the lanes of ROW i
//      (any column), need not be normalised, must be non-zero.
// Out: lambda, unit eigenvector on the lanes of row i (same layout), rounds used.
// All 64 lanes call it.
// ---------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ void wave_row_sum8(double& a, double& b) {
  a += dpp_f64<kDppQuadXor1, 0xF>(a, a);
  b += dpp_f64<kDppQuadXor1, 0xF>(b, b);
  a += dpp_f64<kDppQuadXor2, 0xF>(a, a);
  b += dpp_f64<kDppQuadXor2, 0xF>(b, b);
  a += dpp_f64<kDppRowHalfMirror, 0xF>(a, a);
  b += dpp_f64<kDppRowHalfMirror, 0xF>(b, b);
}

// y = M x: x given per row (component i on row i); the product needs x_j on lane (i, j)
template <int D>
__device__ __forceinline__ void wave_matvec(double mre, double mim, double xr, double xi, LaneIJ c,
                                            bool valid, double& yr, double& yi) {
  const double xjr = lane_get(xr, ij_lane(c.j, 0)), xji = lane_get(xi, ij_lane(c.j, 0));
  yr = valid ? mre * xjr - mim * xji : 0.0;
  yi = valid ? mre * xji + mim * xjr : 0.0;
  wave_row_sum8<D>(yr, yi);  // every lane of row i now holds y_i
}

// sum over the eight rows of the 8 x 8 lane grid (lanes with the same column index), result on
// every lane: three exchange steps instead of the six of wave_sum.  For values that are replicated
// along their row (vector components): sum_i v_i.
__device__ __forceinline__ double wave_colsum(double v) {
  v += dpp_f64<kDppRowRor8, 0xF>(v, v);  // rows i, i ^ 1
  double a = v, b = v;
  swap16_f64(a, b);
  v = a + b;                             // i ^ 2
  a = v;
  b = v;
  swap32_f64(a, b);
  return a + b;                          // i ^ 4
}

constexpr int kDominantMaxRounds = 8;

template <int D>
__device__ __forceinline__ bool wave_dominant_eigenpair(double are, double aim, LaneIJ c,
                                                        double& xr, double& xi, double& lambda,
                                                        int& rounds) {
  const bool valid = c.i < D && c.j < D;
  if (!valid) {
    are = 0.0;
    aim = 0.0;
  }
  if (c.i >= D) {
    xr = 0.0;
    xi = 0.0;
  }
  // scale of the matrix (trace = sum of the eigenvalues >= lambda_max >= trace / D)
  const double tr = wave_sum((valid && c.i == c.j) ? are : 0.0);
  if (!(tr > 0.0) || !(tr < 1.79e308)) return false;
  // residual at which the pair is accepted.  4e-15 tr (round 4) sat within a factor six of the
  // rounding floor of the residual itself (D eps tr): one call in four paid a second inversion
  // (~2 500 cycles) to move a residual of 5e-15 tr below it (profiles/r06_d_watson_phases.txt).
  // 1e-12 tr bounds the eigenvector error by 1e-12 tr / gap; the eigenvalue is second order.
  const double tol = 1e-12 * tr;
  {
    const double n2 = wave_colsum(xr * xr + xi * xi);  // x is replicated along its rows, 0 beyond D
    if (!(n2 > 0.0)) return false;
    const double rn = fast_rsqrt(n2);
    xr *= rn;
    xi *= rn;
  }
  bool certified = false;
  for (rounds = 0; rounds < kDominantMaxRounds; ++rounds) {
    double yr, yi;
    wave_matvec<D>(are, aim, xr, xi, c, valid, yr, yi);
    const double rho = wave_colsum(xr * yr + xi * yi);  // Re x^H C x (x unit)
    const double rr = yr - rho * xr, ri = yi - rho * xi;
    // |r| from a Newton-refined reciprocal square root (~1 ulp; it only places the shift and is
    // compared with the tolerance): the IEEE square root was ~30 dependent instructions, twice per call
    const double r2 = wave_colsum(rr * rr + ri * ri);
    const double res = (r2 > 1e-280 && r2 < 1e280) ? r2 * fast_rsqrt(r2) : sqrt(r2);
    lambda = rho;
    if (!(res < 1.79e308)) return false;
    if (res <= tol && certified) return true;
    // shift above the eigenvalue nearest to rho; the guard keeps the system safely definite
    // once the residual is at rounding level
    const double sigma = rho + 1.01 * res + 64.0 * tol;
    double mre = valid ? ((c.i == c.j) ? sigma - are : -are) : ((c.i == c.j) ? 1.0 : 0.0);
    double mim = valid ? -aim : 0.0;
    ScaledReal det;
    if (wave_hpd_inverse<D>(mre, mim, c, det) != 0) return false;  // an eigenvalue above sigma
    certified = true;  // lambda_max < sigma = rho + 1.01 res + guard
    if (res <= tol) return true;
    // applications of the explicit inverse: each contracts the error by (sigma - l1)/(sigma - l2)
    // ~ the error itself.  Two bring a late EM iteration (mode moved by < 1e-7) to rounding
    // level; an early one (1e-3) needs four -- two more matrix-vector products (~150 cycles
    // each) where a second round would pay a second inversion (~2 500 cycles at D = 6).
    const int reps = (res > 1e-7 * tr) ? 4 : 2;
    for (int rep = 0; rep < reps; ++rep) {
      wave_matvec<D>(mre, mim, xr, xi, c, valid, yr, yi);
      const double n2 = wave_colsum(yr * yr + yi * yi);
      if (!(n2 > 0.0) || !(n2 < 1.79e308)) return false;
      const double rn = fast_rsqrt(n2);
      xr = yr * rn;
      xi = yi * rn;
    }
  }
  return false;
}

// rank of eigenvalue j among the D eigenvalues (ascending, ties by index):
// column j of V belongs at sorted position rank.  lam = value on lane (j,j)
// already broadcast down the column (every lane holds lambda of ITS column j).
template <int D>
__device__ __forceinline__ int wave_sort_rank(double lam_col, LaneIJ c) {
  int rank = 0;
#pragma unroll
  for (int m = 0; m < D; ++m) {
    double lm = lane_get(lam_col, ij_lane(0, m));  // row 0 holds every column's value
    rank += (lm < lam_col || (lm == lam_col && m < c.j)) ? 1 : 0;
  }
  return rank;
}

// ---------------------------------------------------------------------------
// LU with partial pivoting applied to [A | B] (B has M <= D columns, stored in
// lanes j < M of the second matrix).  Returns X = A^-1 B in (xre, xim) and
// sets `singular` when an exactly-zero pivot is met (LAPACK zgesv INFO > 0,
// which numpy turns into LinAlgError("Singular matrix")).
// ---------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ bool wave_lu_solve(double are, double aim, double bre,
                                              double bim, LaneIJ c, double& xre,
                                              double& xim) {
  bool singular = false;
#pragma unroll
  for (int p = 0; p < D; ++p) {
    // pivot search in column p, rows >= p (cabs1-like: |re| + |im| as LAPACK izamax)
    int piv = p;
    double best = -1.0;
#pragma unroll
    for (int r = p; r < D; ++r) {
      double mr = fabs(lane_get(are, ij_lane(r, p))) + fabs(lane_get(aim, ij_lane(r, p)));
      if (mr > best) {
        best = mr;
        piv = r;
      }
    }
    if (!(best > 0.0)) singular = true;
    // swap rows p and piv of A and B
    if (piv != p) {
      int src = (c.i == p) ? piv : ((c.i == piv) ? p : c.i);
      double t0 = lane_get(are, ij_lane(src, c.j)), t1 = lane_get(aim, ij_lane(src, c.j));
      double t2 = lane_get(bre, ij_lane(src, c.j)), t3 = lane_get(bim, ij_lane(src, c.j));
      are = t0;
      aim = t1;
      bre = t2;
      bim = t3;
    }
    double pr = lane_get(are, ij_lane(p, p)), pim = lane_get(aim, ij_lane(p, p));
    double den = pr * pr + pim * pim;
    double ir = singular ? 0.0 : pr / den, ii = singular ? 0.0 : -pim / den;  // 1/pivot
    // multiplier l_ip = a_ip / pivot for rows i > p
    double ar = lane_get(are, ij_lane(c.i, p)), ai = lane_get(aim, ij_lane(c.i, p));
    double lr = ar * ir - ai * ii, li = ar * ii + ai * ir;
    double ur = lane_get(are, ij_lane(p, c.j)), ui = lane_get(aim, ij_lane(p, c.j));
    double vr = lane_get(bre, ij_lane(p, c.j)), vi = lane_get(bim, ij_lane(p, c.j));
    if (c.i > p) {
      are -= lr * ur - li * ui;
      aim -= lr * ui + li * ur;
      bre -= lr * vr - li * vi;
      bim -= lr * vi + li * vr;
    }
  }
  // back substitution on the upper triangle: rows from D-1 down to 0
#pragma unroll
  for (int p = D - 1; p >= 0; --p) {
    double pr = lane_get(are, ij_lane(p, p)), pim = lane_get(aim, ij_lane(p, p));
    double den = pr * pr + pim * pim;
    double ir = (den > 0.0) ? pr / den : 0.0, ii = (den > 0.0) ? -pim / den : 0.0;
    if (c.i == p) {  // x_p = b_p / u_pp
      double nr = bre * ir - bim * ii, ni = bre * ii + bim * ir;
      bre = nr;
      bim = ni;
    }
    double ur = lane_get(are, ij_lane(c.i, p)), ui = lane_get(aim, ij_lane(c.i, p));
    double xr = lane_get(bre, ij_lane(p, c.j)), xi = lane_get(bim, ij_lane(p, c.j));
    if (c.i < p) {
      bre -= ur * xr - ui * xi;
      bim -= ur * xi + ui * xr;
    }
  }
  xre = bre;
  xim = bim;
  return singular;
}


// ---------------------------------------------------------------------------
// Minimum-norm least squares X = A^+ B for an (exactly) singular A, the
// on-device counterpart of the numpy.linalg.lstsq fallback in
// math/solve.py:95-114 and extraction/beamformer.py:251-256.
// Hermitian A (every PSD matrix): A^+ = V diag(1/lambda_i if |lambda_i| >
// D*eps*max|lambda| else 0) V^H -- the same cut-off numpy's lstsq applies to
// the singular values.  General A: X = (A^H A)^+ A^H B.
// ---------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ void wave_pinv_solve(double are, double aim, double bre,
                                                double bim, LaneIJ c, double& xre,
                                                double& xim) {
  const bool valid = c.i < D && c.j < D;
  if (!valid) {
    are = 0.0;
    aim = 0.0;
  }
  if (c.i >= D) {
    bre = 0.0;
    bim = 0.0;
  }
  double tre, tim;
  wave_adjoint(are, aim, c, tre, tim);
  const double dif = wave_sum((are - tre) * (are - tre) + (aim - tim) * (aim - tim));
  const double nrm = wave_sum(are * are + aim * aim);
  const bool herm = dif <= 1e-28 * nrm;
  double gre, gim, rre, rim;
  if (herm) {
    gre = 0.5 * (are + tre);
    gim = 0.5 * (aim + tim);
    rre = bre;
    rim = bim;
  } else {
    wave_matmul<D>(tre, tim, are, aim, c, gre, gim);  // A^H A
    wave_matmul<D>(tre, tim, bre, bim, c, rre, rim);  // A^H B
    double gtr, gti;
    wave_adjoint(gre, gim, c, gtr, gti);
    gre = 0.5 * (gre + gtr);
    gim = 0.5 * (gim + gti);
  }
  double vre, vim;
  wave_jacobi_heev<D>(gre, gim, c, vre, vim);
  double lam = lane_get(gre, ij_lane(c.j, c.j));
  double lmax = wave_max((c.i == 0 && c.j < D) ? fabs(lam) : 0.0);
  double thr = (double)D * 2.220446049250313e-16 * lmax;
  double inv = (fabs(lam) > thr && c.j < D) ? 1.0 / lam : 0.0;  // per column index
  double vhr, vhi, yre, yim;
  wave_adjoint(vre, vim, c, vhr, vhi);
  wave_matmul<D>(vhr, vhi, rre, rim, c, yre, yim);  // V^H R
  double sc = lane_get(inv, ij_lane(0, c.i));       // scale row e by 1/lambda_e
  yre *= sc;
  yim *= sc;
  if (!valid) {
    vre = 0.0;
    vim = 0.0;
  }
  wave_matmul<D>(vre, vim, yre, yim, c, xre, xim);  // V (...)
}

}  // namespace pbbss
