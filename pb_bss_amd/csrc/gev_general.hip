// General (non-Hermitian) generalized eigenproblem  A v = lambda B v  for small complex
// pencils: the `use_eig=True` path of the reference's GEV beamformer
//   extraction/beamformer.py:352-358  -> cythonized/c_eig.pyx:14-123 (LAPACK zggev), or
//   extraction/beamformer.py:367-411  -> scipy.linalg.eig(a, b)       (the Python fallback)
// Both return, per frequency, the UNIT-2-NORM right eigenvector belonging to
// numpy.argmax(eigenvalues) -- for complex eigenvalues numpy orders by real part, then by
// imaginary part.  Nothing is assumed about A and B beyond B being invertible: they need not be
// Hermitian or definite ("It crashes less often, but really hides, when you have matrices which
// are far from Hermitian", beamformer.py:313-316).
//
// One THREAD per pencil (this is the rarely taken robust path, 10^2..10^3 pencils of 2..32
// sensors: a serial textbook algorithm per thread, all pencils in parallel):
//   1. M = B^-1 A            LU with partial pivoting (singular B -> status SINGULAR)
//   2. Hessenberg reduction  stabilised elementary transformations
//   3. all eigenvalues       complex QR iteration with Wilkinson shifts and deflation
//                            (no convergence in 60 sweeps per eigenvalue -> status EIG_NOCONV,
//                             the .pyx's "The QZ iteration failed")
//   4. eigenvector of the selected eigenvalue by inverse iteration on M, unit 2-norm
// Matrices live in per-thread scratch (dynamic indexing); D <= DMAX in {8, 16, 32}.
#include <hip/hip_runtime.h>
#include <cmath>
#include "beamform.hpp"

namespace pbbss {
namespace {

struct Cx {
  double r, i;
};
__device__ __forceinline__ Cx cx(double r, double i = 0.0) { return {r, i}; }
__device__ __forceinline__ Cx operator+(Cx a, Cx b) { return {a.r + b.r, a.i + b.i}; }
__device__ __forceinline__ Cx operator-(Cx a, Cx b) { return {a.r - b.r, a.i - b.i}; }
__device__ __forceinline__ Cx operator*(Cx a, Cx b) {
  return {a.r * b.r - a.i * b.i, a.r * b.i + a.i * b.r};
}
__device__ __forceinline__ Cx conj(Cx a) { return {a.r, -a.i}; }
__device__ __forceinline__ double abs1(Cx a) { return fabs(a.r) + fabs(a.i); }  // LAPACK cabs1
__device__ __forceinline__ double abs2(Cx a) { return hypot(a.r, a.i); }
__device__ __forceinline__ Cx operator/(Cx a, Cx b) {  // Smith's algorithm
  if (fabs(b.r) >= fabs(b.i)) {
    const double t = b.i / b.r, d = b.r + b.i * t;
    return {(a.r + a.i * t) / d, (a.i - a.r * t) / d};
  }
  const double t = b.r / b.i, d = b.r * t + b.i;
  return {(a.r * t + a.i) / d, (a.i * t - a.r) / d};
}
__device__ __forceinline__ Cx csqrt(Cx a) {
  const double m = abs2(a);
  if (m == 0.0) return {0.0, 0.0};
  double sr = sqrt(0.5 * (m + fabs(a.r)));
  double si = 0.5 * a.i / sr;
  if (a.r >= 0.0) return {sr, si};
  return {fabs(si), (a.i >= 0.0) ? sr : -sr};
}

// In-place LU with partial pivoting of the n x n matrix m (row stride LD); piv receives the row
// permutation.  Returns false when a pivot is exactly zero.
template <int LD>
__device__ bool lu_factor(Cx* m, int* piv, int n) {
  bool ok = true;
  for (int p = 0; p < n; ++p) {
    int best = p;
    double bv = abs1(m[p * LD + p]);
    for (int r = p + 1; r < n; ++r) {
      const double v = abs1(m[r * LD + p]);
      if (v > bv) {
        bv = v;
        best = r;
      }
    }
    piv[p] = best;
    if (best != p) {
      for (int c = 0; c < n; ++c) {
        const Cx t = m[p * LD + c];
        m[p * LD + c] = m[best * LD + c];
        m[best * LD + c] = t;
      }
    }
    if (!(bv > 0.0)) {
      ok = false;
      continue;
    }
    const Cx inv = cx(1.0) / m[p * LD + p];
    for (int r = p + 1; r < n; ++r) {
      const Cx l = m[r * LD + p] * inv;
      m[r * LD + p] = l;
      for (int c = p + 1; c < n; ++c) m[r * LD + c] = m[r * LD + c] - l * m[p * LD + c];
    }
  }
  return ok;
}

// Solve (LU) x = b in place for one right-hand side.
template <int LD>
__device__ void lu_solve(const Cx* lu, const int* piv, int n, Cx* x) {
  for (int p = 0; p < n; ++p) {
    const Cx t = x[p];
    x[p] = x[piv[p]];
    x[piv[p]] = t;
  }
  for (int r = 1; r < n; ++r) {
    Cx s = x[r];
    for (int c = 0; c < r; ++c) s = s - lu[r * LD + c] * x[c];
    x[r] = s;
  }
  for (int r = n - 1; r >= 0; --r) {
    Cx s = x[r];
    for (int c = r + 1; c < n; ++c) s = s - lu[r * LD + c] * x[c];
    x[r] = s / lu[r * LD + r];
  }
}

// Reduce the general matrix h to upper Hessenberg form by stabilised elementary similarity
// transformations (row / column interchanges + eliminations); eigenvalues are preserved.
template <int LD>
__device__ void to_hessenberg(Cx* h, int n) {
  for (int m = 1; m < n - 1; ++m) {
    int best = m;
    double bv = 0.0;
    for (int r = m; r < n; ++r) {
      const double v = abs1(h[r * LD + m - 1]);
      if (v > bv) {
        bv = v;
        best = r;
      }
    }
    if (bv == 0.0) continue;
    if (best != m) {
      for (int c = m - 1; c < n; ++c) {
        const Cx t = h[best * LD + c];
        h[best * LD + c] = h[m * LD + c];
        h[m * LD + c] = t;
      }
      for (int r = 0; r < n; ++r) {
        const Cx t = h[r * LD + best];
        h[r * LD + best] = h[r * LD + m];
        h[r * LD + m] = t;
      }
    }
    const Cx x = h[m * LD + m - 1];
    for (int r = m + 1; r < n; ++r) {
      Cx y = h[r * LD + m - 1];
      if (y.r == 0.0 && y.i == 0.0) continue;
      y = y / x;
      h[r * LD + m - 1] = cx(0.0);
      for (int c = m; c < n; ++c) h[r * LD + c] = h[r * LD + c] - y * h[m * LD + c];
      for (int q = 0; q < n; ++q) h[q * LD + m] = h[q * LD + m] + y * h[q * LD + r];
    }
  }
}

// All eigenvalues of an upper Hessenberg matrix by the explicitly shifted complex QR iteration
// (Givens rotations, Wilkinson shift, deflation at negligible subdiagonals).  h is destroyed.
// Returns false if some eigenvalue did not converge.
template <int LD>
__device__ bool hessenberg_eigenvalues(Cx* h, int n, Cx* lam, Cx* rot_c_s) {
  const double eps = 2.220446049250313e-16;
  int hi = n - 1;
  int its = 0;
  while (hi >= 0) {
    // smallest l such that h[l..hi] is an unreduced block
    int l = hi;
    while (l > 0) {
      const double sub = abs1(h[l * LD + l - 1]);
      const double nb = abs1(h[(l - 1) * LD + l - 1]) + abs1(h[l * LD + l]);
      if (sub <= eps * (nb > 0.0 ? nb : 1.0)) {
        h[l * LD + l - 1] = cx(0.0);
        break;
      }
      --l;
    }
    if (l == hi) {  // 1 x 1 block: converged
      lam[hi] = h[hi * LD + hi];
      --hi;
      its = 0;
      continue;
    }
    if (++its > 60) return false;
    // Wilkinson shift: eigenvalue of the trailing 2 x 2 closer to h[hi][hi]
    Cx mu;
    {
      const Cx a = h[(hi - 1) * LD + hi - 1], b = h[(hi - 1) * LD + hi];
      const Cx c = h[hi * LD + hi - 1], d = h[hi * LD + hi];
      const Cx half_tr = (a + d) * cx(0.5);
      const Cx dm = (a - d) * cx(0.5);
      const Cx disc = csqrt(dm * dm + b * c);
      const Cx e1 = half_tr + disc, e2 = half_tr - disc;
      mu = (abs2(e1 - d) <= abs2(e2 - d)) ? e1 : e2;
      if (its == 10 || its == 20) mu = mu + cx(abs1(c), 0.0);  // exceptional shift against stagnation
    }
    for (int k = l; k <= hi; ++k) h[k * LD + k] = h[k * LD + k] - mu;
    // QR sweep on the active block: H - mu I = Q R (rows), then R Q (columns)
    for (int k = l; k < hi; ++k) {
      const Cx a = h[k * LD + k], b = h[(k + 1) * LD + k];
      const double nrm = hypot(abs2(a), abs2(b));
      Cx c, s;  // G = [conj(c) conj(s); -s c] with c = a/nrm, s = b/nrm  ->  G [a; b] = [nrm; 0]
      if (nrm == 0.0) {
        c = cx(1.0);
        s = cx(0.0);
      } else {
        c = cx(a.r / nrm, a.i / nrm);
        s = cx(b.r / nrm, b.i / nrm);
      }
      rot_c_s[2 * k] = c;
      rot_c_s[2 * k + 1] = s;
      for (int col = k; col <= hi; ++col) {  // eigenvalues only: the active window suffices
        const Cx x = h[k * LD + col], y = h[(k + 1) * LD + col];
        h[k * LD + col] = conj(c) * x + conj(s) * y;
        h[(k + 1) * LD + col] = c * y - s * x;
      }
    }
    for (int k = l; k < hi; ++k) {  // right-multiply by G_k^H: columns k, k+1
      const Cx c = rot_c_s[2 * k], s = rot_c_s[2 * k + 1];
      const int rmax = (k + 2 <= hi) ? k + 2 : hi;
      for (int r = l; r <= rmax; ++r) {
        const Cx x = h[r * LD + k], y = h[r * LD + k + 1];
        h[r * LD + k] = x * c + y * s;
        h[r * LD + k + 1] = y * conj(c) - x * conj(s);
      }
    }
    for (int k = l; k <= hi; ++k) h[k * LD + k] = h[k * LD + k] + mu;
  }
  return true;
}

template <int DMAX>
__global__ void __launch_bounds__(64) gev_general_kernel(const double* __restrict__ target,
                                                         const double* __restrict__ noise,
                                                         int64_t N, int D, double* __restrict__ out_w,
                                                         double* __restrict__ out_lambda,
                                                         int32_t* __restrict__ out_status) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  constexpr int LD = DMAX;
  Cx m[DMAX * DMAX];     // B^-1 A
  Cx wk[DMAX * DMAX];    // LU of B, then Hessenberg / QR workspace, then LU of (M - lambda I)
  Cx lam[DMAX], x[DMAX], rot[2 * DMAX];
  int piv[DMAX];
  int st = 0;
  const double* a = target + n * D * D * 2;
  const double* b = noise + n * D * D * 2;
  for (int r = 0; r < D; ++r)
    for (int c = 0; c < D; ++c) {
      wk[r * LD + c] = cx(b[(r * D + c) * 2], b[(r * D + c) * 2 + 1]);
      m[r * LD + c] = cx(a[(r * D + c) * 2], a[(r * D + c) * 2 + 1]);
    }
  if (!lu_factor<LD>(wk, piv, D)) st |= PBBSS_ST_SINGULAR;
  if (st == 0) {
    // M = B^-1 A column by column
    for (int c = 0; c < D; ++c) {
      for (int r = 0; r < D; ++r) x[r] = m[r * LD + c];
      lu_solve<LD>(wk, piv, D, x);
      for (int r = 0; r < D; ++r) m[r * LD + c] = x[r];
    }
    for (int r = 0; r < D; ++r)
      for (int c = 0; c < D; ++c) wk[r * LD + c] = m[r * LD + c];
    to_hessenberg<LD>(wk, D);
    if (!hessenberg_eigenvalues<LD>(wk, D, lam, rot)) st |= PBBSS_ST_EIG_NOCONV;
  }
  Cx best = cx(0.0);
  if (st == 0) {
    // numpy.argmax over complex values: larger real part, ties by the imaginary part, the
    // first maximum wins; NaNs are avoided by the finite check below
    int bi = 0;
    for (int k = 1; k < D; ++k)
      if (lam[k].r > lam[bi].r || (lam[k].r == lam[bi].r && lam[k].i > lam[bi].i)) bi = k;
    best = lam[bi];
    if (!(isfinite(best.r) && isfinite(best.i))) st |= PBBSS_ST_NONFINITE;
  }
  if (st == 0) {
    // inverse iteration: (M - (lambda + delta) I) x_{j+1} = x_j ; a tiny relative offset keeps the
    // shifted matrix numerically regular
    double scale = 0.0;
    for (int r = 0; r < D; ++r)
      for (int c = 0; c < D; ++c) scale = fmax(scale, abs1(m[r * LD + c]));
    const double delta = 64.0 * 2.220446049250313e-16 * (scale > 0.0 ? scale : 1.0);
    for (int r = 0; r < D; ++r)
      for (int c = 0; c < D; ++c)
        wk[r * LD + c] = (r == c) ? m[r * LD + c] - best - cx(delta, delta) : m[r * LD + c];
    const bool regular = lu_factor<LD>(wk, piv, D);
    if (!regular) {
      // exactly singular shifted matrix: nudge the zero pivots (the null vector is what we want)
      for (int r = 0; r < D; ++r)
        if (abs1(wk[r * LD + r]) == 0.0) wk[r * LD + r] = cx(delta);
    }
    for (int r = 0; r < D; ++r) x[r] = cx(1.0 + 0.25 * r, 0.5 - 0.125 * r);
    for (int itv = 0; itv < 3; ++itv) {
      lu_solve<LD>(wk, piv, D, x);
      double big = 0.0;
      for (int r = 0; r < D; ++r) big = fmax(big, abs1(x[r]));
      if (!(big > 0.0) || !isfinite(big)) {
        st |= PBBSS_ST_NONFINITE;
        break;
      }
      double n2 = 0.0;
      for (int r = 0; r < D; ++r) {
        x[r] = cx(x[r].r / big, x[r].i / big);
        n2 += x[r].r * x[r].r + x[r].i * x[r].i;
      }
      const double inv = 1.0 / sqrt(n2);
      for (int r = 0; r < D; ++r) x[r] = cx(x[r].r * inv, x[r].i * inv);
    }
  }
  for (int r = 0; r < D; ++r) {
    out_w[(n * D + r) * 2] = (st == 0) ? x[r].r : 0.0;
    out_w[(n * D + r) * 2 + 1] = (st == 0) ? x[r].i : 0.0;
  }
  if (out_lambda) {
    out_lambda[n * 2] = best.r;
    out_lambda[n * 2 + 1] = best.i;
  }
  if (out_status) out_status[n] = st;
}

}  // namespace

int launch_gev_general(const double* target, const double* noise, int64_t N, int D, double* out_w,
                       double* out_lambda, int32_t* out_status, hipStream_t stream) {
  if (D < 1 || D > 32) return PBBSS_ERR_UNSUPPORTED;
  const unsigned grid = (unsigned)((N + 63) / 64);
  if (D <= 8) {
    hipLaunchKernelGGL(gev_general_kernel<8>, dim3(grid), dim3(64), 0, stream, target, noise, N, D,
                       out_w, out_lambda, out_status);
  } else if (D <= 16) {
    hipLaunchKernelGGL(gev_general_kernel<16>, dim3(grid), dim3(64), 0, stream, target, noise, N,
                       D, out_w, out_lambda, out_status);
  } else {
    hipLaunchKernelGGL(gev_general_kernel<32>, dim3(grid), dim3(64), 0, stream, target, noise, N,
                       D, out_w, out_lambda, out_status);
  }
  return hipGetLastError() == hipSuccess ? PBBSS_OK : PBBSS_ERR_HIP;
}

}  // namespace pbbss
