// Workgroup-cooperative dense linear algebra on ONE complex matrix per 256-thread workgroup,
// matrices in LDS (row stride LD, interleaved re/im), sizes up to 32 x 32: the building blocks
// of the generic-size (9 <= D <= 32) kernels.  Counterpart of wave_la.hpp (one matrix per
// wavefront in registers, D <= 8).  Every routine must be called by all threads of the
// workgroup; they synchronise internally and leave the workgroup synchronised on return.
#pragma once
#include "pbbss_dev.hpp"

namespace pbbss {

constexpr int kGenThreads = 256;
constexpr int kGenWaves = kGenThreads / kWave;
constexpr int kGenMaxSweeps = 30;
constexpr double kGenJacobiTol = 1e-29;       // off-diagonal Frobenius^2 / total Frobenius^2
constexpr double kGenJacobiTolLoose = 1e-24;  // accepted if the sweep budget runs out

// NT: threads of the workgroup (kGenThreads everywhere except the eigensolver kernel, which
// runs up to 1024 threads per matrix); red holds NT / 64 doubles
template <int NT = kGenThreads>
__device__ __forceinline__ double gen_block_sum(double v, double* red, int tid) {
  v = wave_sum(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int w = 0; w < NT / kWave; ++w) s += red[w];
  return s;
}

// Scratch of the Jacobi solver: A2, V2 (LD*LD*2 doubles each), rot (3 LD), red (threads / 64),
// part (LD ints)
struct GenJacobiScratch {
  double* A2;
  double* V2;
  double* rot;
  double* red;
  int* part;
};

// Hermitian A (D x D) -> diagonal (eigenvalues on the diagonal of A, unsorted), V = eigenvectors
// in columns.  Parallel cyclic Jacobi, circle-method pairing, ping-pong buffers.
// Returns the number of sweeps, or -1 if not converged (even to the loose tolerance).
template <int NT = kGenThreads>
__device__ inline int lds_jacobi_heev(double* A, double* V, const GenJacobiScratch& S, int D,
                                      int LD, int tid) {
  const int N = D + (D & 1);
  for (int e = tid; e < LD * LD; e += NT) {
    const int i = e / LD, j = e - i * LD;
    V[e * 2] = (i == j) ? 1.0 : 0.0;
    V[e * 2 + 1] = 0.0;
    if (i >= D || j >= D) {
      A[e * 2] = 0.0;
      A[e * 2 + 1] = 0.0;
    } else if (i == j) {
      A[e * 2 + 1] = 0.0;
    }
  }
  __syncthreads();
  double fro2 = 0.0;
  for (int e = tid; e < LD * LD; e += NT) fro2 += A[e * 2] * A[e * 2] + A[e * 2 + 1] * A[e * 2 + 1];
  fro2 = gen_block_sum<NT>(fro2, S.red, tid);
  if (!(fro2 > 0.0) || !isfinite(fro2)) return 0;
  int sweeps = -1;
  double* Vc = V;      // current eigenvector estimate
  double* Vn = S.V2;   // next
  for (int e = tid; e < LD * LD * 2; e += NT) S.V2[e] = V[e];  // identity padding in both
  __syncthreads();
  // the entries of this thread in the two update passes (fixed for the whole solve: the
  // index arithmetic -- two integer divisions per entry and pass -- is hoisted out of the
  // ~200 rounds)
  constexpr int kOwn = (32 * 32 + NT - 1) / NT;  // D <= 32 (generic.hpp: kGenMaxD)
  int own_i[kOwn], own_j[kOwn];
#pragma unroll
  for (int m = 0; m < kOwn; ++m) {
    const int e0 = tid + m * NT;
    own_i[m] = (e0 < D * D) ? e0 / D : -1;
    own_j[m] = (e0 < D * D) ? e0 - own_i[m] * D : 0;
  }
  for (int sweep = 0; sweep < kGenMaxSweeps; ++sweep) {
    double off2 = 0.0;
    for (int e = tid; e < LD * LD; e += NT) {
      const int i = e / LD, j = e - i * LD;
      if (i != j) off2 += A[e * 2] * A[e * 2] + A[e * 2 + 1] * A[e * 2 + 1];
    }
    off2 = gen_block_sum<NT>(off2, S.red, tid);
    if (off2 <= kGenJacobiTol * fro2) {
      sweeps = sweep;
      break;
    }
    for (int r = 0; r < N - 1; ++r) {
      if (tid < LD) {
        const int x = tid;
        int y;
        if (x >= N) y = x;
        else if (x == N - 1) y = r;
        else if (x == r) y = N - 1;
        else {
          y = (2 * r - x) % (N - 1);
          if (y < 0) y += N - 1;
        }
        S.part[x] = y;
        const int p = x < y ? x : y, q = x < y ? y : x;
        if (x == p) {
          double c = 1.0, sr = 0.0, si = 0.0;
          if (p != q && q < D) {
            const double app = A[(p * LD + p) * 2], aqq = A[(q * LD + q) * 2];
            const double xr = A[(p * LD + q) * 2], xi = A[(p * LD + q) * 2 + 1];
            const double g2 = xr * xr + xi * xi;
            const double d = aqq - app;
            const double h2 = fma(d, d, 4.0 * g2);
            if (g2 > 0.0 && h2 < 1.79e308) {  // rotation as in wave_jacobi_heev (two rsqrt)
              const double rh = fast_rsqrt(h2);
              const double c2 = fma(0.5 * fabs(d), rh, 0.5);
              const double rc = fast_rsqrt(c2);
              c = c2 * rc;
              const double ig = ((d < 0.0) ? -rh : rh) * rc;
              sr = xr * ig;
              si = xi * ig;
            }
          }
          S.rot[p * 3] = c;
          S.rot[p * 3 + 1] = sr;
          S.rot[p * 3 + 2] = si;
          if (q != p && q < LD) {
            S.rot[q * 3] = c;
            S.rot[q * 3 + 1] = sr;
            S.rot[q * 3 + 2] = si;
          }
        }
      }
      __syncthreads();
      // rows: B = J^H A   (only the D x D block is live; the padding stays zero)
#pragma unroll
      for (int m = 0; m < kOwn; ++m) {
        const int i = own_i[m], j = own_j[m];
        if (i < 0) continue;
        const int e = i * LD + j;
        const int pi = S.part[i];
        const double c = S.rot[i * 3], sr = S.rot[i * 3 + 1], si = S.rot[i * 3 + 2];
        const double ar = A[e * 2], ai = A[e * 2 + 1];
        const double orr = A[(pi * LD + j) * 2], oii = A[(pi * LD + j) * 2 + 1];
        double nr = ar, ni = ai;
        if (pi >= D) {
          // partner is the dummy index of an odd-sized tournament: identity
        } else if (i < pi) {
          nr = c * ar - (sr * orr - si * oii);
          ni = c * ai - (sr * oii + si * orr);
        } else if (i > pi) {
          nr = c * ar + (sr * orr + si * oii);
          ni = c * ai + (sr * oii - si * orr);
        }
        S.A2[e * 2] = nr;
        S.A2[e * 2 + 1] = ni;
      }
      __syncthreads();
      // columns: A' = B J, V' = V J   (V ping-pongs between the two buffers)
#pragma unroll
      for (int m = 0; m < kOwn; ++m) {
        const int i = own_i[m], j = own_j[m];
        if (i < 0) continue;
        const int e = i * LD + j;
        const int pj = S.part[j];
        const double c = S.rot[j * 3], sr = S.rot[j * 3 + 1], si = S.rot[j * 3 + 2];
        const double ar = S.A2[e * 2], ai = S.A2[e * 2 + 1];
        const double orr = S.A2[(i * LD + pj) * 2], oii = S.A2[(i * LD + pj) * 2 + 1];
        const double vr = Vc[e * 2], vi = Vc[e * 2 + 1];
        const double wr = Vc[(i * LD + pj) * 2], wi = Vc[(i * LD + pj) * 2 + 1];
        double nr = ar, ni = ai, xr = vr, xi = vi;
        if (pj >= D) {
          // dummy partner: identity (and A2 / V hold nothing meaningful in that column)
        } else if (j < pj) {
          nr = c * ar - (sr * orr + si * oii);
          ni = c * ai - (sr * oii - si * orr);
          xr = c * vr - (sr * wr + si * wi);
          xi = c * vi - (sr * wi - si * wr);
        } else if (j > pj) {
          nr = c * ar + (sr * orr - si * oii);
          ni = c * ai + (sr * oii + si * orr);
          xr = c * vr + (sr * wr - si * wi);
          xi = c * vi + (sr * wi + si * wr);
        }
        if (i == j) ni = 0.0;
        A[e * 2] = nr;
        A[e * 2 + 1] = ni;
        Vn[e * 2] = xr;
        Vn[e * 2 + 1] = xi;
      }
      __syncthreads();
      double* tswap = Vc;
      Vc = Vn;
      Vn = tswap;
    }
  }
  if (sweeps < 0) {
    double off2 = 0.0;
    for (int e = tid; e < LD * LD; e += NT) {
      const int i = e / LD, j = e - i * LD;
      if (i != j) off2 += A[e * 2] * A[e * 2] + A[e * 2 + 1] * A[e * 2 + 1];
    }
    off2 = gen_block_sum<NT>(off2, S.red, tid);
    if (off2 <= kGenJacobiTolLoose * fro2) sweeps = kGenMaxSweeps;
  }
  if (Vc != V) {  // an odd number of rounds left the result in the scratch buffer
    for (int e = tid; e < LD * LD * 2; e += NT) V[e] = Vc[e];
  }
  __syncthreads();
  return sweeps;
}

// ------------------------------------------------------------------ Hermitian eigensolver (one wavefront)
// LDS operations of ONE wavefront execute in program order, so a store by one lane is seen by a
// later load of another lane without a barrier; this only keeps the compiler from reordering.
__device__ __forceinline__ void wave_lds_fence() { __asm__ volatile("" ::: "memory"); }

// Hermitian A (n x n, LDS, complex row stride LD, full matrix stored) -> eigenvalue `dreg` and
// eigenvector (xre, xim)[0 .. n) of index `lane` (unsorted; lanes >= n hold zeros), by
// Householder tridiagonalisation (LAPACK zhetd2's recurrences, reflectors H_k = I - tau_k v_k
// v_k^H kept below the subdiagonal of A), implicit-shift QL on (d, e) with the rotations
// accumulated in a REAL matrix Z (EISPACK tql2 / Numerical Recipes tqli) and the
// back-transformation V = H_0 ... H_{n-3} Z with one eigenvector per lane in registers.
// Called by ONE wavefront (all 64 lanes, lane = threadIdx & 63); A is destroyed.
// Scratch: Zt (n columns of LDz >= n doubles, Zt[col * LDz + row], set to the IDENTITY by the
// caller), dv, ev (DP + 1 doubles), tauv, vbuf, wbuf (2 DP doubles each).  solve = false returns
// the diagonal with V = I.
// Returns true when the QL iteration did not converge.
template <int DP>
__device__ inline bool wave_heev_ql(double* A, const int LD, double* Zt, const int LDz, double* dv,
                                    double* ev,
                                    double* tauv, double* vbuf, double* wbuf, const int n,
                                    const int lane, const bool solve, double& dreg_out,
                                    double (&xre)[DP], double (&xim)[DP]) {
  if (lane < n) {  // a zero / non-finite matrix is returned as its diagonal with V = I
    dv[lane] = A[(lane * LD + lane) * 2];
    ev[lane] = 0.0;
  }
  wave_lds_fence();
  // ---- Householder tridiagonalisation: T = Q^H A Q, Q = H_0 H_1 ... H_{n-2}
  for (int k = 0; solve && k + 1 < n; ++k) {
    const bool below = lane > k && lane < n;        // rows k+1 .. n-1
    const bool tail = lane > k + 1 && lane < n;     // rows k+2 .. n-1
    const double xr = below ? A[(lane * LD + k) * 2] : 0.0;
    const double xi = below ? A[(lane * LD + k) * 2 + 1] : 0.0;
    const double ar = A[((k + 1) * LD + k) * 2], ai = A[((k + 1) * LD + k) * 2 + 1];
    const double xn2 = wave_sum(tail ? xr * xr + xi * xi : 0.0);
    double tr_ = 0.0, ti_ = 0.0, beta = ar, vr = 0.0, vi = 0.0;
    if (!(xn2 == 0.0 && ai == 0.0)) {  // zlarfg
      beta = -copysign(sqrt(ar * ar + ai * ai + xn2), ar);
      tr_ = (beta - ar) / beta;
      ti_ = -ai / beta;
      const double dr = ar - beta, di = ai, den = dr * dr + di * di;
      const double sr = dr / den, si = -di / den;  // 1 / (alpha - beta)
      vr = tail ? xr * sr - xi * si : 0.0;
      vi = tail ? xr * si + xi * sr : 0.0;
    }
    if (lane == k + 1) {
      vr = 1.0;
      vi = 0.0;
    }
    if (lane < DP) {
      vbuf[lane * 2] = vr;
      vbuf[lane * 2 + 1] = vi;
    }
    if (tail) {  // keep the reflector for the back-transformation
      A[(lane * LD + k) * 2] = vr;
      A[(lane * LD + k) * 2 + 1] = vi;
    }
    if (lane == 0) {
      ev[k] = beta;
      dv[k] = A[(k * LD + k) * 2];
      tauv[k * 2] = tr_;
      tauv[k * 2 + 1] = ti_;
    }
    wave_lds_fence();
    if (tr_ != 0.0 || ti_ != 0.0) {  // uniform
      // p = tau A22 v, row `lane` through the Hermitian mirror A[lane][j] = conj(A[j][lane])
      double pr = 0.0, pi = 0.0;
      for (int j = k + 1; j < n; ++j) {
        const double ajr = below ? A[(j * LD + lane) * 2] : 0.0;
        const double aji = below ? A[(j * LD + lane) * 2 + 1] : 0.0;
        const double ur = vbuf[j * 2], ui = vbuf[j * 2 + 1];
        pr += ajr * ur + aji * ui;   // conj(a) * u
        pi += ajr * ui - aji * ur;
      }
      {
        const double qr = tr_ * pr - ti_ * pi, qi = tr_ * pi + ti_ * pr;
        pr = qr;
        pi = qi;
      }
      // w = p - 1/2 tau (p^H v) v
      double dr = below ? pr * vr + pi * vi : 0.0;  // conj(p) v
      double di = below ? pr * vi - pi * vr : 0.0;
      dr = wave_sum(dr);
      di = wave_sum(di);
      const double hr = -0.5 * (tr_ * dr - ti_ * di), hi = -0.5 * (tr_ * di + ti_ * dr);
      const double wr = pr + hr * vr - hi * vi, wi = pi + hr * vi + hi * vr;
      if (lane < DP) {
        wbuf[lane * 2] = below ? wr : 0.0;
        wbuf[lane * 2 + 1] = below ? wi : 0.0;
      }
      wave_lds_fence();
      // A22 -= v w^H + w v^H, lane = column
      if (below) {
        for (int r = k + 1; r < n; ++r) {
          const double ur = vbuf[r * 2], ui = vbuf[r * 2 + 1];
          const double sr = wbuf[r * 2], si = wbuf[r * 2 + 1];
          double* a = A + (r * LD + lane) * 2;
          // v_r conj(w_c) + w_r conj(v_c)
          a[0] -= ur * wr + ui * wi + sr * vr + si * vi;
          a[1] -= ui * wr - ur * wi + si * vr - sr * vi;
        }
      }
      wave_lds_fence();
    }
  }
  if (lane == 0) {
    dv[n - 1] = A[((n - 1) * LD + n - 1) * 2];
    ev[n - 1] = 0.0;
  }
  wave_lds_fence();
  // ---- implicit-shift QL on (d, e).  d and e live in REGISTERS (lane i holds d[i], e[i]); the
  // scalar recurrences run redundantly in all lanes on values fetched with v_readlane (the
  // index is wave-uniform), so the serial chain never waits for LDS; lane = row of Z applies
  // the rotations (one LDS read and one write per rotation: the column shared by two
  // consecutive rotations stays in a register).
  double dreg = (lane < n) ? dv[lane] : 0.0, ereg = (lane < n) ? ev[lane] : 0.0;
  bool failed = false;
  for (int l = 0; solve && l < n; ++l) {
    int iter = 0;
    for (;;) {
      // first negligible subdiagonal at or after l: lane m tests e[m], one ballot
      int m;
      {
        const double dnext = __shfl_down(dreg, 1);
        bool neg = false;
        if (lane >= l && lane + 1 < n) {
          const double dd = fabs(dreg) + fabs(dnext);
          neg = (fabs(ereg) + dd == dd);
        }
        const unsigned long long mask = __ballot(neg);
        m = __builtin_amdgcn_readfirstlane(mask ? (int)__builtin_ctzll(mask) : n - 1);
      }
      if (m == l) break;
      if (++iter > 60) {
        failed = true;
        break;
      }
      const double dl = lane_bcast_const(dreg, l), el = lane_bcast_const(ereg, l);
      double gg = (lane_bcast_const(dreg, l + 1) - dl) / (2.0 * el);
      double r = sqrt(gg * gg + 1.0);
      gg = lane_bcast_const(dreg, m) - dl + el / (gg + copysign(r, gg));
      double s = 1.0, c = 1.0, p = 0.0;
      const int zl = (lane < n) ? lane : 0;
      double zc = Zt[m * LDz + zl];        // column i + 1 of Z, carried
      double zi = Zt[(m - 1) * LDz + zl];  // column i, fetched one rotation ahead of its use
      int i = m - 1;
      bool under = false;
      double ei = lane_bcast_const(ereg, i), di = lane_bcast_const(dreg, i);
      double di1 = lane_bcast_const(dreg, m);
      for (; i >= l; --i) {
        // operands of the NEXT rotation first: they do not depend on the chain below (entries
        // below i + 1 are not written during this sweep)
        const int inx = (i > 0) ? i - 1 : 0;
        const double ei_n = lane_bcast_const(ereg, inx), di_n = lane_bcast_const(dreg, inx);
        const double znext = Zt[inx * LDz + zl];
        const double f = s * ei, b = c * ei;
        // r = hypot(f, g), s = f / r, c = g / r through ONE Newton-refined reciprocal square
        // root, branch-free: this scalar recurrence is the serial chain of the whole solver
        const double h2 = fma(f, f, gg * gg);
        const double rinv = fast_rsqrt(fmax(h2, 1e-300));
        r = h2 * rinv;
        if (lane == i + 1) ereg = r;
        if (r == 0.0) {  // recover from underflow (f = g = 0): restart this eigenvalue
          if (lane == i + 1) dreg = di1 - p;
          if (lane == m) ereg = 0.0;
          under = true;
          break;
        }
        s = f * rinv;
        c = gg * rinv;
        gg = di1 - p;
        r = fma(di - gg, s, 2.0 * c * b);
        p = s * r;
        if (lane == i + 1) dreg = gg + p;
        gg = fma(c, r, -b);
        if (lane < n) Zt[(i + 1) * LDz + lane] = fma(s, zi, c * zc);
        zc = fma(c, zi, -s * zc);
        zi = znext;
        di1 = di;
        di = di_n;
        ei = ei_n;
      }
      if (lane < n) Zt[(under ? i + 1 : l) * LDz + lane] = zc;  // the carried column goes home
      if (under) continue;
      if (lane == l) {
        dreg -= p;
        ereg = gg;
      }
      if (lane == m) ereg = 0.0;
    }
    if (failed) break;
  }
  wave_lds_fence();
  // ---- eigenvector `lane`: x = H_0 ... H_{n-3} Z[:, lane] (H_{n-2} has v = e_{n-1}: a phase)
#pragma unroll
  for (int r = 0; r < DP; ++r) {
    xre[r] = (r < n && lane < n) ? Zt[lane * LDz + r] : 0.0;
    xim[r] = 0.0;
  }
  for (int k = n - 2; solve && k >= 0; --k) {
    const double tr_ = tauv[k * 2], ti_ = tauv[k * 2 + 1];
    if (tr_ == 0.0 && ti_ == 0.0) continue;  // uniform
    // v_k: 1 at k + 1, A[r][k] below, 0 above; s = v^H x
    // the reflector is fetched ONCE, lane r taking v_k[r] (1 at k + 1, A[r][k] below, 0
    // above), and handed to the FMAs through v_readlane (r is a compile-time constant of the
    // unrolled loops, the value arrives as an SGPR operand): no LDS round trip per row, no
    // operand arrays next to the 2 DP registers of the eigenvector
    const bool on = lane > k + 1 && lane < n;
    const double are = A[((on ? lane : n - 1) * LD + k) * 2];
    const double aim = A[((on ? lane : n - 1) * LD + k) * 2 + 1];
    const double vre = (lane == k + 1) ? 1.0 : (on ? are : 0.0);
    const double vim = on ? aim : 0.0;
    double sr = 0.0, si = 0.0;
#pragma unroll
    for (int r = 1; r < DP; ++r) {
      const double ur = lane_bcast_const(vre, r), ui = lane_bcast_const(vim, r);
      sr += ur * xre[r] + ui * xim[r];
      si += ur * xim[r] - ui * xre[r];
    }
    const double qr = tr_ * sr - ti_ * si, qi = tr_ * si + ti_ * sr;  // tau (v^H x)
#pragma unroll
    for (int r = 1; r < DP; ++r) {
      const double ur = lane_bcast_const(vre, r), ui = lane_bcast_const(vim, r);
      xre[r] -= ur * qr - ui * qi;
      xim[r] -= ur * qi + ui * qr;
    }
  }
  dreg_out = dreg;
  return failed;
}

// C = A B (all n x n in LDS, stride LD); C must not alias A or B
__device__ inline void lds_matmul(const double* A, const double* Bm, double* C, int n, int LD,
                                  int tid, bool adjoint_a = false, bool adjoint_b = false) {
  for (int e = tid; e < n * n; e += kGenThreads) {
    const int i = e / n, j = e - i * n;
    double sr = 0.0, si = 0.0;
    for (int m = 0; m < n; ++m) {
      double ar, ai, br, bi;
      if (adjoint_a) {
        ar = A[(m * LD + i) * 2];
        ai = -A[(m * LD + i) * 2 + 1];
      } else {
        ar = A[(i * LD + m) * 2];
        ai = A[(i * LD + m) * 2 + 1];
      }
      if (adjoint_b) {
        br = Bm[(j * LD + m) * 2];
        bi = -Bm[(j * LD + m) * 2 + 1];
      } else {
        br = Bm[(m * LD + j) * 2];
        bi = Bm[(m * LD + j) * 2 + 1];
      }
      sr += ar * br - ai * bi;
      si += ar * bi + ai * br;
    }
    C[(i * LD + j) * 2] = sr;
    C[(i * LD + j) * 2 + 1] = si;
  }
  __syncthreads();
}

// LU with partial pivoting (LAPACK zgesv: pivot = first maximum of |re| + |im|) applied to
// [A | B], B has M columns (stride LD).  On return B = A^-1 B.  `flag` (LDS int) is set to 1
// when an exactly-zero pivot is met (numpy: LinAlgError "Singular matrix"); A and B are then
// partially overwritten.
__device__ inline bool lds_lu_solve(double* A, double* Bm, int D, int M, int LD, int* flag,
                                    int tid) {
  if (tid == 0) flag[0] = 0;
  __syncthreads();
  for (int p = 0; p < D; ++p) {
    if (tid == 0) {
      int best = p;
      double bv = fabs(A[(p * LD + p) * 2]) + fabs(A[(p * LD + p) * 2 + 1]);
      for (int i = p + 1; i < D; ++i) {
        const double v = fabs(A[(i * LD + p) * 2]) + fabs(A[(i * LD + p) * 2 + 1]);
        if (v > bv) {
          bv = v;
          best = i;
        }
      }
      flag[1] = best;
      if (!(bv > 0.0)) flag[0] = 1;
    }
    __syncthreads();
    if (flag[0]) return true;
    const int piv = flag[1];
    if (piv != p) {
      for (int c = tid; c < D + M; c += kGenThreads) {
        double* a = (c < D) ? A + (p * LD + c) * 2 : Bm + (p * LD + (c - D)) * 2;
        double* b = (c < D) ? A + (piv * LD + c) * 2 : Bm + (piv * LD + (c - D)) * 2;
        const double tr = a[0], ti = a[1];
        a[0] = b[0];
        a[1] = b[1];
        b[0] = tr;
        b[1] = ti;
      }
    }
    __syncthreads();
    const double pr = A[(p * LD + p) * 2], pi = A[(p * LD + p) * 2 + 1];
    const double pd = pr * pr + pi * pi;
    const int rows = D - 1 - p, cols = (D - 1 - p) + M;
    // factors first (column p is read by every thread of a row before it is left behind)
    for (int e = tid; e < rows * cols; e += kGenThreads) {
      const int i = p + 1 + e / cols, cc = e - (e / cols) * cols;
      const double ar = A[(i * LD + p) * 2], ai = A[(i * LD + p) * 2 + 1];
      const double fr = (ar * pr + ai * pi) / pd, fi = (ai * pr - ar * pi) / pd;  // a_ip / a_pp
      double* dst;
      const double* src;
      if (cc < D - 1 - p) {
        dst = A + (i * LD + p + 1 + cc) * 2;
        src = A + (p * LD + p + 1 + cc) * 2;
      } else {
        dst = Bm + (i * LD + (cc - (D - 1 - p))) * 2;
        src = Bm + (p * LD + (cc - (D - 1 - p))) * 2;
      }
      dst[0] -= fr * src[0] - fi * src[1];
      dst[1] -= fr * src[1] + fi * src[0];
    }
    __syncthreads();
  }
  // back substitution, one thread per right-hand side
  for (int m = tid; m < M; m += kGenThreads) {
    for (int p = D - 1; p >= 0; --p) {
      double sr = Bm[(p * LD + m) * 2], si = Bm[(p * LD + m) * 2 + 1];
      for (int c = p + 1; c < D; ++c) {
        const double ar = A[(p * LD + c) * 2], ai = A[(p * LD + c) * 2 + 1];
        const double xr = Bm[(c * LD + m) * 2], xi = Bm[(c * LD + m) * 2 + 1];
        sr -= ar * xr - ai * xi;
        si -= ar * xi + ai * xr;
      }
      const double pr = A[(p * LD + p) * 2], pi = A[(p * LD + p) * 2 + 1];
      const double pd = pr * pr + pi * pi;
      Bm[(p * LD + m) * 2] = (sr * pr + si * pi) / pd;
      Bm[(p * LD + m) * 2 + 1] = (si * pr - sr * pi) / pd;
    }
  }
  __syncthreads();
  return false;
}

// Minimum-norm least squares X = A^+ B for an exactly singular A (numpy.linalg.lstsq with
// rcond=None as used by math/solve.py:95-114): Hermitian A through its eigendecomposition
// (cut-off D eps max|lambda|), general A through (A^H A)^+ A^H B.  A, Bm hold the ORIGINAL
// system; G, V, T are LD*LD*2 work matrices; the solution is written to Bm.
__device__ inline void lds_pinv_solve(const double* A, double* Bm, double* G, double* V, double* T,
                                      const GenJacobiScratch& S, int D, int M, int LD, int tid) {
  double dif = 0.0, nrm = 0.0;
  for (int e = tid; e < D * D; e += kGenThreads) {
    const int i = e / D, j = e - i * D;
    const double ar = A[(i * LD + j) * 2], ai = A[(i * LD + j) * 2 + 1];
    const double tr = A[(j * LD + i) * 2], ti = -A[(j * LD + i) * 2 + 1];
    dif += (ar - tr) * (ar - tr) + (ai - ti) * (ai - ti);
    nrm += ar * ar + ai * ai;
  }
  dif = gen_block_sum(dif, S.red, tid);
  nrm = gen_block_sum(nrm, S.red, tid);
  const bool herm = dif <= 1e-28 * nrm;
  // G = Hermitian system matrix, T = right-hand side
  for (int e = tid; e < LD * LD; e += kGenThreads) {
    const int i = e / LD, j = e - i * LD;
    double gr = 0.0, gi = 0.0;
    if (i < D && j < D) {
      if (herm) {
        gr = 0.5 * (A[(i * LD + j) * 2] + A[(j * LD + i) * 2]);
        gi = 0.5 * (A[(i * LD + j) * 2 + 1] - A[(j * LD + i) * 2 + 1]);
      } else {
        for (int m = 0; m < D; ++m) {  // (A^H A)_ij
          const double ar = A[(m * LD + i) * 2], ai = -A[(m * LD + i) * 2 + 1];
          const double br = A[(m * LD + j) * 2], bi = A[(m * LD + j) * 2 + 1];
          gr += ar * br - ai * bi;
          gi += ar * bi + ai * br;
        }
      }
    }
    G[e * 2] = gr;
    G[e * 2 + 1] = gi;
    double tr = 0.0, ti = 0.0;
    if (i < D && j < M) {
      if (herm) {
        tr = Bm[(i * LD + j) * 2];
        ti = Bm[(i * LD + j) * 2 + 1];
      } else {
        for (int m = 0; m < D; ++m) {  // (A^H B)_ij
          const double ar = A[(m * LD + i) * 2], ai = -A[(m * LD + i) * 2 + 1];
          const double br = Bm[(m * LD + j) * 2], bi = Bm[(m * LD + j) * 2 + 1];
          tr += ar * br - ai * bi;
          ti += ar * bi + ai * br;
        }
      }
    }
    T[e * 2] = tr;
    T[e * 2 + 1] = ti;
  }
  __syncthreads();
  if (!herm) {  // symmetrise the rounding noise of A^H A
    for (int e = tid; e < D * D; e += kGenThreads) {
      const int i = e / D, j = e - i * D;
      if (i < j) {
        const double mr = 0.5 * (G[(i * LD + j) * 2] + G[(j * LD + i) * 2]);
        const double mi = 0.5 * (G[(i * LD + j) * 2 + 1] - G[(j * LD + i) * 2 + 1]);
        G[(i * LD + j) * 2] = mr;
        G[(i * LD + j) * 2 + 1] = mi;
        G[(j * LD + i) * 2] = mr;
        G[(j * LD + i) * 2 + 1] = -mi;
      }
    }
    __syncthreads();
  }
  lds_jacobi_heev(G, V, S, D, LD, tid);
  double lmax = 0.0;
  for (int e = 0; e < D; ++e) lmax = fmax(lmax, fabs(G[(e * LD + e) * 2]));
  const double thr = (double)D * 2.220446049250313e-16 * lmax;
  // Y = diag(1/lambda) V^H T  (into S.A2), X = V Y (into Bm)
  for (int e = tid; e < D * M; e += kGenThreads) {
    const int r = e / M, j = e - r * M;
    const double lam = G[(r * LD + r) * 2];
    const double inv = (fabs(lam) > thr) ? 1.0 / lam : 0.0;
    double sr = 0.0, si = 0.0;
    for (int m = 0; m < D; ++m) {
      const double vr = V[(m * LD + r) * 2], vi = -V[(m * LD + r) * 2 + 1];
      const double tr = T[(m * LD + j) * 2], ti = T[(m * LD + j) * 2 + 1];
      sr += vr * tr - vi * ti;
      si += vr * ti + vi * tr;
    }
    S.A2[(r * LD + j) * 2] = sr * inv;
    S.A2[(r * LD + j) * 2 + 1] = si * inv;
  }
  __syncthreads();
  for (int e = tid; e < D * M; e += kGenThreads) {
    const int i = e / M, j = e - i * M;
    double sr = 0.0, si = 0.0;
    for (int r = 0; r < D; ++r) {
      const double vr = V[(i * LD + r) * 2], vi = V[(i * LD + r) * 2 + 1];
      const double yr = S.A2[(r * LD + j) * 2], yi = S.A2[(r * LD + j) * 2 + 1];
      sr += vr * yr - vi * yi;
      si += vr * yi + vi * yr;
    }
    Bm[(i * LD + j) * 2] = sr;
    Bm[(i * LD + j) * 2 + 1] = si;
  }
  __syncthreads();
}

// In-place Cholesky A = L L^H of a Hermitian matrix (lower triangle on return, strict upper
// zeroed).  Returns 0 or 1 + index of the first non-positive / non-finite pivot (zpotrf INFO).
__device__ inline int lds_cholesky(double* A, int D, int LD, int* flag, int tid) {
  if (tid == 0) flag[0] = 0;
  __syncthreads();
  for (int p = 0; p < D; ++p) {
    if (tid == 0) {
      const double d = A[(p * LD + p) * 2];
      if (!(d > 0.0) || !isfinite(d)) flag[0] = p + 1;
      else A[(p * LD + p) * 2] = sqrt(d);
      A[(p * LD + p) * 2 + 1] = 0.0;
    }
    __syncthreads();
    if (flag[0]) return flag[0];
    const double lpp = A[(p * LD + p) * 2];
    for (int i = p + 1 + tid; i < D; i += kGenThreads) {
      A[(i * LD + p) * 2] /= lpp;
      A[(i * LD + p) * 2 + 1] /= lpp;
    }
    __syncthreads();
    // trailing update of the lower triangle: a_ij -= l_ip conj(l_jp), i >= j > p
    const int n = D - 1 - p;
    for (int e = tid; e < n * n; e += kGenThreads) {
      const int i = p + 1 + e / n, j = p + 1 + e - (e / n) * n;
      if (i >= j) {
        const double ar = A[(i * LD + p) * 2], ai = A[(i * LD + p) * 2 + 1];
        const double br = A[(j * LD + p) * 2], bi = -A[(j * LD + p) * 2 + 1];
        A[(i * LD + j) * 2] -= ar * br - ai * bi;
        A[(i * LD + j) * 2 + 1] -= ar * bi + ai * br;
      }
    }
    __syncthreads();
  }
  for (int e = tid; e < D * D; e += kGenThreads) {
    const int i = e / D, j = e - i * D;
    if (i < j) {
      A[(i * LD + j) * 2] = 0.0;
      A[(i * LD + j) * 2 + 1] = 0.0;
    }
  }
  __syncthreads();
  return 0;
}

}  // namespace pbbss
