// Generic-size (9 <= D <= 32 sensors) beamformer kernels: one 256-thread workgroup per
// frequency bin, matrices in LDS (generic_dev.hpp).  Same semantics, status bits and reference
// lines as the one-wavefront-per-matrix kernels of beamform.hip:
//   solve         math/solve.py:20-114 (LU, least-squares fallback for exactly singular matrices)
//   mvdr_souden   extraction/beamformer.py:627-698 (modes 1/2: wMWF :701-753)
//   mvdr          :230-260        ban  :459-488        gev  :292-411 / get_gev_vector.pyx:42-150
#include "generic_bf.hpp"
#include <cmath>
#include "generic_dev.hpp"

namespace pbbss {
namespace {

constexpr int LD = 32;  // LDS row stride of every matrix here
constexpr int kMat = LD * LD * 2;

struct Work {
  double *A, *B, *A0, *B0, *G, *V, *T2;
  GenJacobiScratch S;
  int* flag;
};
constexpr size_t kWorkBytes = (size_t)(9 * kMat + 3 * LD + kGenWaves) * sizeof(double) + (LD + 4) * sizeof(int);

__device__ inline Work carve(char* smem) {
  Work w;
  double* p = reinterpret_cast<double*>(smem);
  w.A = p;  p += kMat;
  w.B = p;  p += kMat;
  w.A0 = p; p += kMat;
  w.B0 = p; p += kMat;
  w.G = p;  p += kMat;
  w.V = p;  p += kMat;
  w.T2 = p; p += kMat;
  w.S.A2 = p; p += kMat;
  w.S.V2 = p; p += kMat;
  w.S.rot = p; p += 3 * LD;
  w.S.red = p; p += kGenWaves;
  w.S.part = reinterpret_cast<int*>(p);
  w.flag = w.S.part + LD;
  return w;
}

__device__ inline void load_mat(const double* src, int64_t n, int D, int cols, double* dst, int tid) {
  // (n, D, cols) row-major complex -> LDS stride LD (rest zero)
  for (int e = tid; e < LD * LD; e += kGenThreads) {
    const int i = e / LD, j = e - i * LD;
    double re = 0.0, im = 0.0;
    if (i < D && j < cols) {
      const double* p = src + (((size_t)n * D + i) * cols + j) * 2;
      re = p[0];
      im = p[1];
    }
    dst[e * 2] = re;
    dst[e * 2 + 1] = im;
  }
}

__device__ inline void copy_mat(const double* src, double* dst, int tid) {
  for (int e = tid; e < kMat; e += kGenThreads) dst[e] = src[e];
}

// X = A^-1 B with the reference's fallback; A0/B0 keep the original system for the fallback
__device__ inline bool stable_solve(const Work& w, int D, int M, int tid) {
  copy_mat(w.A, w.A0, tid);
  copy_mat(w.B, w.B0, tid);
  __syncthreads();
  const bool sing = lds_lu_solve(w.A, w.B, D, M, LD, w.flag, tid);
  if (sing) {
    copy_mat(w.B0, w.B, tid);
    __syncthreads();
    lds_pinv_solve(w.A0, w.B, w.G, w.V, w.T2, w.S, D, M, LD, tid);
  }
  return sing;
}

__global__ void __launch_bounds__(kGenThreads)
    gen_solve_kernel(const double* A, const double* Bm, int D, int M, double* out, int32_t* status) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const Work w = carve(smem);
  const int tid = threadIdx.x;
  const int64_t n = blockIdx.x;
  load_mat(A, n, D, D, w.A, tid);
  load_mat(Bm, n, D, M, w.B, tid);
  __syncthreads();
  const bool sing = stable_solve(w, D, M, tid);
  for (int e = tid; e < D * M; e += kGenThreads) {
    const int i = e / M, j = e - i * M;
    out[(((size_t)n * D + i) * M + j) * 2] = w.B[(i * LD + j) * 2];
    out[(((size_t)n * D + i) * M + j) * 2 + 1] = w.B[(i * LD + j) * 2 + 1];
  }
  if (status && tid == 0) status[n] = sing ? PBBSS_ST_SINGULAR : 0;
}

// mode 0 MVDR-Souden, 1 wMWF (eps = mu), 2 wMWF 'frequency_dependent' -- as beamform.hip
__global__ void __launch_bounds__(kGenThreads)
    gen_souden_kernel(const double* target, const double* noise, int D, double eps, int mode,
                      double* out_mat, double* snr_num, double* snr_den, int32_t* status) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const Work w = carve(smem);
  const int tid = threadIdx.x;
  const int64_t n = blockIdx.x;
  load_mat(noise, n, D, D, w.A, tid);
  load_mat(target, n, D, D, w.B, tid);
  __syncthreads();
  const bool sing = stable_solve(w, D, D, tid);  // G = noise^-1 target in w.B; originals in A0/B0
  double trr = 0.0, tri = 0.0;
  for (int d = 0; d < D; ++d) {
    trr += w.B[(d * LD + d) * 2];
    tri += w.B[(d * LD + d) * 2 + 1];
  }
  double dr, di;
  if (mode == 0) {
    dr = fmax(trr, eps);
    di = 0.0;
  } else if (mode == 1) {
    dr = eps + trr;
    di = tri;
  } else {
    const double pr = w.B0[0], pi = w.B0[1];  // phi_x1x1
    const double zr = pr * trr - pi * tri, zi = pr * tri + pi * trr;
    const double mag = sqrt(sqrt(zr * zr + zi * zi)), ang = 0.5 * atan2(zi, zr);
    dr = mag * cos(ang);
    di = mag * sin(ang);
  }
  const double den = dr * dr + di * di;
  __syncthreads();
  for (int e = tid; e < D * D; e += kGenThreads) {
    const int i = e / D, j = e - i * D;
    const double gr = w.B[(i * LD + j) * 2], gi = w.B[(i * LD + j) * 2 + 1];
    double nr, ni;
    if (mode == 0) {
      nr = gr / dr;
      ni = gi / dr;
    } else {
      nr = (gr * dr + gi * di) / den;
      ni = (gi * dr - gr * di) / den;
    }
    w.G[(i * LD + j) * 2] = nr;
    w.G[(i * LD + j) * 2 + 1] = ni;
    out_mat[(((size_t)n * D + i) * D + j) * 2] = nr;
    out_mat[(((size_t)n * D + i) * D + j) * 2 + 1] = ni;
  }
  __syncthreads();
  if (snr_num || snr_den) {
    // num_r = sum_{d,e} conj(mat_dr) target_de mat_er ; den_r with noise (beamformer.py:616-620)
    lds_matmul(w.B0, w.G, w.V, D, LD, tid);   // target * mat
    lds_matmul(w.A0, w.G, w.T2, D, LD, tid);  // noise * mat
    for (int r = tid; r < D; r += kGenThreads) {
      double pr = 0.0, pi = 0.0, qr = 0.0, qi = 0.0;
      for (int d = 0; d < D; ++d) {
        const double mr = w.G[(d * LD + r) * 2], mi = w.G[(d * LD + r) * 2 + 1];
        const double ar = w.V[(d * LD + r) * 2], ai = w.V[(d * LD + r) * 2 + 1];
        const double br = w.T2[(d * LD + r) * 2], bi = w.T2[(d * LD + r) * 2 + 1];
        pr += mr * ar + mi * ai;
        pi += mr * ai - mi * ar;
        qr += mr * br + mi * bi;
        qi += mr * bi - mi * br;
      }
      if (snr_num) {
        snr_num[((size_t)n * D + r) * 2] = pr;
        snr_num[((size_t)n * D + r) * 2 + 1] = pi;
      }
      if (snr_den) {
        snr_den[((size_t)n * D + r) * 2] = qr;
        snr_den[((size_t)n * D + r) * 2 + 1] = qi;
      }
    }
  }
  if (status && tid == 0) status[n] = sing ? PBBSS_ST_SINGULAR : 0;
}

__global__ void __launch_bounds__(kGenThreads)
    gen_mvdr_kernel(const double* atf, const double* noise, int D, double* out_w, int32_t* status) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const Work w = carve(smem);
  const int tid = threadIdx.x;
  const int64_t n = blockIdx.x;
  load_mat(noise, n, D, D, w.G, tid);
  load_mat(atf, n, D, 1, w.B, tid);
  __syncthreads();
  for (int e = tid; e < LD * LD; e += kGenThreads) {  // hermitise (:246-248)
    const int i = e / LD, j = e - i * LD;
    w.A[e * 2] = 0.5 * (w.G[(i * LD + j) * 2] + w.G[(j * LD + i) * 2]);
    w.A[e * 2 + 1] = 0.5 * (w.G[(i * LD + j) * 2 + 1] - w.G[(j * LD + i) * 2 + 1]);
  }
  __syncthreads();
  const bool sing = stable_solve(w, D, 1, tid);
  double dr = 0.0, di = 0.0;  // h^H x
  for (int d = 0; d < D; ++d) {
    const double hr = w.B0[(d * LD) * 2], hi = w.B0[(d * LD) * 2 + 1];
    const double xr = w.B[(d * LD) * 2], xi = w.B[(d * LD) * 2 + 1];
    dr += hr * xr + hi * xi;
    di += hr * xi - hi * xr;
  }
  const double den = dr * dr + di * di;
  for (int d = tid; d < D; d += kGenThreads) {
    const double xr = w.B[(d * LD) * 2], xi = w.B[(d * LD) * 2 + 1];
    out_w[((size_t)n * D + d) * 2] = (xr * dr + xi * di) / den;
    out_w[((size_t)n * D + d) * 2 + 1] = (xi * dr - xr * di) / den;
  }
  if (status && tid == 0) status[n] = sing ? PBBSS_ST_SINGULAR : 0;
}

// one thread per bin: w * | sqrt(w^H P P w) / |w^H P w| |
__global__ void gen_ban_kernel(const double* wv, const double* noise, int64_t N, int D, double* out) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const double* P = noise + (size_t)n * D * D * 2;
  const double* w = wv + (size_t)n * D * 2;
  double nr = 0.0, ni = 0.0, dr = 0.0, di = 0.0;
  for (int b = 0; b < D; ++b) {
    double ur = 0.0, ui = 0.0, vr = 0.0, vi = 0.0;  // u_b = (P w)_b, v_b = (w^H P)_b
    for (int c = 0; c < D; ++c) {
      const double pr = P[(b * D + c) * 2], pi = P[(b * D + c) * 2 + 1];
      ur += pr * w[c * 2] - pi * w[c * 2 + 1];
      ui += pr * w[c * 2 + 1] + pi * w[c * 2];
      const double qr = P[(c * D + b) * 2], qi = P[(c * D + b) * 2 + 1];
      vr += w[c * 2] * qr + w[c * 2 + 1] * qi;
      vi += w[c * 2] * qi - w[c * 2 + 1] * qr;
    }
    nr += vr * ur - vi * ui;
    ni += vr * ui + vi * ur;
    dr += w[b * 2] * ur + w[b * 2 + 1] * ui;
    di += w[b * 2] * ui - w[b * 2 + 1] * ur;
  }
  const double dabs = sqrt(dr * dr + di * di);
  const double scale = (dabs != 0.0) ? sqrt(sqrt(nr * nr + ni * ni)) / dabs : 0.0;
  for (int d = 0; d < D; ++d) {
    out[((size_t)n * D + d) * 2] = w[d * 2] * scale;
    out[((size_t)n * D + d) * 2 + 1] = w[d * 2 + 1] * scale;
  }
}

// principal generalised eigenvector, w^H N w = 1 (zhegvd ITYPE=1): N = L L^H,
// M = L^-1 T L^-H, top eigenvector u of M, w = L^-H u
__global__ void __launch_bounds__(kGenThreads)
    gen_gev_kernel(const double* target, const double* noise, int D, double* out_w, int32_t* status) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const Work w = carve(smem);
  const int tid = threadIdx.x;
  const int64_t n = blockIdx.x;
  int st = 0;
  // Hermitian matrices defined by the UPPER triangle of the stored arrays (what LAPACK reads
  // through the Cython wrapper, cythonized/get_gev_vector.pyx:72-74)
  double bad = 0.0;
  for (int e = tid; e < LD * LD; e += kGenThreads) {
    const int i = e / LD, j = e - i * LD;
    double tr = 0.0, ti = 0.0, nr = 0.0, ni = 0.0;
    if (i < D && j < D) {
      const int lo = i < j ? i : j, hi = i < j ? j : i;
      const double* p = target + (((size_t)n * D + lo) * D + hi) * 2;
      const double* q = noise + (((size_t)n * D + lo) * D + hi) * 2;
      tr = p[0];
      ti = (i == j) ? 0.0 : ((i < j) ? p[1] : -p[1]);
      nr = q[0];
      ni = (i == j) ? 0.0 : ((i < j) ? q[1] : -q[1]);
      if (!isfinite(tr) || !isfinite(ti) || !isfinite(nr) || !isfinite(ni)) bad = 1.0;
    }
    w.B0[e * 2] = tr;
    w.B0[e * 2 + 1] = ti;
    w.A[e * 2] = nr;
    w.A[e * 2 + 1] = ni;
  }
  if (gen_block_sum(bad, w.S.red, tid) > 0.0) st |= PBBSS_ST_NONFINITE;
  __syncthreads();
  const int info = lds_cholesky(w.A, D, LD, w.flag, tid);
  if (info != 0) st |= PBBSS_ST_NOT_POSDEF | (info << 8);
  // X = L^-1: forward substitution on the identity, one thread per column
  for (int e = tid; e < kMat; e += kGenThreads) w.B[e] = 0.0;
  __syncthreads();
  if (info == 0) {
    for (int c = tid; c < D; c += kGenThreads) {
      for (int i = c; i < D; ++i) {
        double sr = (i == c) ? 1.0 : 0.0, si = 0.0;
        for (int m = c; m < i; ++m) {
          const double lr = w.A[(i * LD + m) * 2], li = w.A[(i * LD + m) * 2 + 1];
          const double xr = w.B[(m * LD + c) * 2], xi = w.B[(m * LD + c) * 2 + 1];
          sr -= lr * xr - li * xi;
          si -= lr * xi + li * xr;
        }
        const double d = w.A[(i * LD + i) * 2];
        w.B[(i * LD + c) * 2] = sr / d;
        w.B[(i * LD + c) * 2 + 1] = si / d;
      }
    }
  }
  __syncthreads();
  lds_matmul(w.B, w.B0, w.T2, D, LD, tid);               // X T
  lds_matmul(w.T2, w.B, w.G, D, LD, tid, false, true);   // (X T) X^H
  for (int e = tid; e < D * D; e += kGenThreads) {        // symmetrise rounding noise
    const int i = e / D, j = e - i * D;
    if (i < j) {
      const double mr = 0.5 * (w.G[(i * LD + j) * 2] + w.G[(j * LD + i) * 2]);
      const double mi = 0.5 * (w.G[(i * LD + j) * 2 + 1] - w.G[(j * LD + i) * 2 + 1]);
      w.G[(i * LD + j) * 2] = mr;
      w.G[(i * LD + j) * 2 + 1] = mi;
      w.G[(j * LD + i) * 2] = mr;
      w.G[(j * LD + i) * 2 + 1] = -mi;
    }
  }
  __syncthreads();
  // eigendecomposition of G.  D > 20: tridiagonal QL by the first wavefront (generic_dev.hpp;
  // 1.94 -> 1.71 ms for the 513 bins of a D = 29 chain), only the principal eigenvector is kept
  // and written to column 0 of V.  Smaller matrices: the 256-thread Jacobi is faster here (this
  // kernel holds one workgroup per CU either way: 0.58 vs 0.70 ms at D = 16).
  int col = 0;
  if (D <= 20) {
    if (lds_jacobi_heev(w.G, w.V, w.S, D, LD, tid) < 0) st |= PBBSS_ST_EIG_NOCONV;
    __syncthreads();
    double best = -1.79e308;
    for (int m = 0; m < D; ++m) {
      const double lam = w.G[(m * LD + m) * 2];
      if (lam >= best) {  // ascending order, ties by index: the last maximum
        best = lam;
        col = m;
      }
    }
  } else if (tid < kWave) {
    const int lane = tid;
    double* Zt = w.T2;                  // [D][LD] real
    double* dv = w.T2 + LD * LD;        // the rest of T2 holds the vectors of the solver
    double* ev = dv + LD + 1;
    double* tauv = ev + LD + 1;
    double* vbuf = tauv + 2 * LD;
    double* wbuf = vbuf + 2 * LD;
    for (int e = lane; e < D * LD; e += kWave) Zt[e] = ((e / LD) == (e % LD)) ? 1.0 : 0.0;
    wave_lds_fence();
    double fro2 = 0.0;
    for (int e = lane; e < D * D; e += kWave) {
      const int i = e / D, j = e - i * D;
      fro2 += w.G[(i * LD + j) * 2] * w.G[(i * LD + j) * 2] + w.G[(i * LD + j) * 2 + 1] * w.G[(i * LD + j) * 2 + 1];
    }
    fro2 = wave_sum(fro2);
    double lam, xre[LD], xim[LD];
    if (wave_heev_ql<LD>(w.G, LD, Zt, LD, dv, ev, tauv, vbuf, wbuf, D, lane,
                         (fro2 > 0.0) && isfinite(fro2), lam, xre, xim))
      st |= PBBSS_ST_EIG_NOCONV;
    // ascending order, ties by index: the last maximum
    int top = 0;
    double best = -1.79e308;
    for (int m = 0; m < D; ++m) {
      const double lm = lane_bcast_const(lam, m);
      if (lm >= best) {
        best = lm;
        top = m;
      }
    }
    if (lane == top) {
#pragma unroll
      for (int r = 0; r < LD; ++r) {
        if (r < D) {
          w.V[(r * LD) * 2] = xre[r];
          w.V[(r * LD) * 2 + 1] = xim[r];
        }
      }
    }
  }
  __syncthreads();
  {  // OR of the per-thread status words of the first wavefront into every thread's copy
    __shared__ int sst;
    if (tid == 0) sst = 0;
    __syncthreads();
    if (st) atomicOr(&sst, st);
    __syncthreads();
    st = sst;
  }
  for (int i = tid; i < D; i += kGenThreads) {  // w_i = sum_m conj(X_mi) u_m
    double sr = 0.0, si = 0.0;
    for (int m = 0; m < D; ++m) {
      const double ar = w.B[(m * LD + i) * 2], ai = w.B[(m * LD + i) * 2 + 1];
      const double ur = w.V[(m * LD + col) * 2], ui = w.V[(m * LD + col) * 2 + 1];
      sr += ar * ur + ai * ui;
      si += ar * ui - ai * ur;
    }
    out_w[((size_t)n * D + i) * 2] = sr;
    out_w[((size_t)n * D + i) * 2 + 1] = si;
  }
  if (status && tid == 0) status[n] = st;
}

inline int ok_or_hip() { return hipGetLastError() == hipSuccess ? PBBSS_OK : PBBSS_ERR_HIP; }

template <typename KFN>
int prep(KFN kfn, size_t lds_limit) {
  if (kWorkBytes > lds_limit) return PBBSS_ERR_LDS_CAPACITY;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWorkBytes) != hipSuccess)
    return PBBSS_ERR_HIP;
  return PBBSS_OK;
}

}  // namespace

int launch_gen_solve(const double* A, const double* Bm, int64_t N, int D, int M, double* x,
                     int32_t* st, size_t lds_limit, hipStream_t s) {
  if (D < 2 || D > LD || M < 1 || M > LD) return PBBSS_ERR_UNSUPPORTED;
  int rc = prep(gen_solve_kernel, lds_limit);
  if (rc != PBBSS_OK) return rc;
  hipLaunchKernelGGL(gen_solve_kernel, dim3((unsigned)N), dim3(kGenThreads), kWorkBytes, s, A, Bm,
                     D, M, x, st);
  return ok_or_hip();
}

int launch_gen_souden(const double* t, const double* nn, int64_t N, int D, double eps, int mode,
                      double* mat, double* num, double* den, int32_t* st, size_t lds_limit,
                      hipStream_t s) {
  if (D < 2 || D > LD) return PBBSS_ERR_UNSUPPORTED;
  int rc = prep(gen_souden_kernel, lds_limit);
  if (rc != PBBSS_OK) return rc;
  hipLaunchKernelGGL(gen_souden_kernel, dim3((unsigned)N), dim3(kGenThreads), kWorkBytes, s, t, nn,
                     D, eps, mode, mat, num, den, st);
  return ok_or_hip();
}

int launch_gen_mvdr(const double* atf, const double* nn, int64_t N, int D, double* w, int32_t* st,
                    size_t lds_limit, hipStream_t s) {
  if (D < 2 || D > LD) return PBBSS_ERR_UNSUPPORTED;
  int rc = prep(gen_mvdr_kernel, lds_limit);
  if (rc != PBBSS_OK) return rc;
  hipLaunchKernelGGL(gen_mvdr_kernel, dim3((unsigned)N), dim3(kGenThreads), kWorkBytes, s, atf, nn,
                     D, w, st);
  return ok_or_hip();
}

int launch_gen_ban(const double* w, const double* nn, int64_t N, int D, double* out, hipStream_t s) {
  hipLaunchKernelGGL(gen_ban_kernel, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, s, w, nn, N, D,
                     out);
  return ok_or_hip();
}

int launch_gen_gev(const double* t, const double* nn, int64_t N, int D, double* w, int32_t* st,
                   size_t lds_limit, hipStream_t s) {
  if (D < 2 || D > LD) return PBBSS_ERR_UNSUPPORTED;
  int rc = prep(gen_gev_kernel, lds_limit);
  if (rc != PBBSS_OK) return rc;
  hipLaunchKernelGGL(gen_gev_kernel, dim3((unsigned)N), dim3(kGenThreads), kWorkBytes, s, t, nn, D,
                     w, st);
  return ok_or_hip();
}

}  // namespace pbbss
