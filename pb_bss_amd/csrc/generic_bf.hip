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

// one wavefront per bin, lane = row: w * | sqrt(w^H P P w) / |w^H P w| |  (one thread per bin
// walked D^2 dependent loads: 0.23 ms for the 513 bins of a D = 29 utterance)
__global__ void __launch_bounds__(256)
    gen_ban_kernel(const double* wv, const double* noise, int64_t N, int D, double* out) {
  const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (n >= N) return;
  const double* P = noise + (size_t)n * D * D * 2;
  const double* w = wv + (size_t)n * D * 2;
  const int b = lane < D ? lane : 0;
  double ur = 0.0, ui = 0.0, vr = 0.0, vi = 0.0;  // u_b = (P w)_b, v_b = (w^H P)_b
  for (int c = 0; c < D; ++c) {
    const double wr = w[c * 2], wi = w[c * 2 + 1];
    const double pr = P[(b * D + c) * 2], pi = P[(b * D + c) * 2 + 1];
    ur += pr * wr - pi * wi;
    ui += pr * wi + pi * wr;
    const double qr = P[(c * D + b) * 2], qi = P[(c * D + b) * 2 + 1];
    vr += wr * qr + wi * qi;
    vi += wr * qi - wi * qr;
  }
  const bool on = lane < D;
  const double wbr = w[b * 2], wbi = w[b * 2 + 1];
  double nr = on ? vr * ur - vi * ui : 0.0, ni = on ? vr * ui + vi * ur : 0.0;
  double dr = on ? wbr * ur + wbi * ui : 0.0, di = on ? wbr * ui - wbi * ur : 0.0;
  nr = wave_sum(nr);
  ni = wave_sum(ni);
  dr = wave_sum(dr);
  di = wave_sum(di);
  const double dabs = sqrt(dr * dr + di * di);
  const double scale = (dabs != 0.0) ? sqrt(sqrt(nr * nr + ni * ni)) / dabs : 0.0;
  if (on) {
    out[((size_t)n * D + b) * 2] = wbr * scale;
    out[((size_t)n * D + b) * 2 + 1] = wbi * scale;
  }
}

// principal generalised eigenvector, w^H N w = 1 (zhegvd ITYPE=1): N = L L^H,
// M = L^-1 T L^-H, top eigenvector u of M, w = L^-H u.
// LDS: THREE matrix slots instead of the nine of the common work area, so that three workgroups
// share a compute unit and the 513 pencils of an utterance run in one round (the kernel is a
// chain of latency-bound steps -- Cholesky, substitution, two products, the eigensolver on one
// wavefront --: 1.1 ms at one workgroup per CU, round 2):
//   slot 0  N -> L (Cholesky in place) -> X T (L is dead once X = L^-1 exists) -> solver scratch
//   slot 1  X = L^-1
//   slot 2  T -> M = (X T) X^H (T is dead once X T exists)
// The eigensolver is the tridiagonal QL of generic_dev.hpp on the first wavefront for every size
// (the 256-thread Jacobi needs three more matrix slots); only the principal eigenvector is kept.
constexpr size_t kGevBytes =
    (size_t)(3 * kMat + 2 * LD + kGenWaves) * sizeof(double) + (LD + 4) * sizeof(int);

__global__ void __launch_bounds__(kGenThreads)
    gen_gev_kernel(const double* target, const double* noise, int D, double* out_w, int32_t* status) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* A = reinterpret_cast<double*>(smem);  // slot 0
  double* X = A + kMat;                         // slot 1
  double* B0 = X + kMat;                        // slot 2
  double* T2 = A;                               // X T, later the solver's scratch
  double* G = B0;                               // M
  double* vtop = B0 + kMat;                     // [LD][2] principal eigenvector
  double* red = vtop + 2 * LD;                  // [kGenWaves]
  int* flag = reinterpret_cast<int*>(red + kGenWaves);
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
    B0[e * 2] = tr;
    B0[e * 2 + 1] = ti;
    A[e * 2] = nr;
    A[e * 2 + 1] = ni;
  }
  if (gen_block_sum(bad, red, tid) > 0.0) st |= PBBSS_ST_NONFINITE;
  __syncthreads();
  const int info = lds_cholesky(A, D, LD, flag, tid);
  if (info != 0) st |= PBBSS_ST_NOT_POSDEF | (info << 8);
  // X = L^-1: forward substitution on the identity, one thread per column
  for (int e = tid; e < kMat; e += kGenThreads) X[e] = 0.0;
  __syncthreads();
  if (info == 0) {
    for (int c = tid; c < D; c += kGenThreads) {
      for (int i = c; i < D; ++i) {
        double sr = (i == c) ? 1.0 : 0.0, si = 0.0;
        for (int m = c; m < i; ++m) {
          const double lr = A[(i * LD + m) * 2], li = A[(i * LD + m) * 2 + 1];
          const double xr = X[(m * LD + c) * 2], xi = X[(m * LD + c) * 2 + 1];
          sr -= lr * xr - li * xi;
          si -= lr * xi + li * xr;
        }
        const double d = A[(i * LD + i) * 2];
        X[(i * LD + c) * 2] = sr / d;
        X[(i * LD + c) * 2 + 1] = si / d;
      }
    }
  }
  __syncthreads();
  lds_matmul(X, B0, T2, D, LD, tid);               // X T   (overwrites L)
  lds_matmul(T2, X, G, D, LD, tid, false, true);   // (X T) X^H   (overwrites T)
  for (int e = tid; e < D * D; e += kGenThreads) {  // symmetrise rounding noise
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
  if (tid < kWave) {
    const int lane = tid;
    double* Zt = T2;                  // [D][LD] real
    double* dv = T2 + LD * LD;        // the rest of the slot holds the vectors of the solver
    double* ev = dv + LD + 1;
    double* tauv = ev + LD + 1;
    double* vbuf = tauv + 2 * LD;
    double* wbuf = vbuf + 2 * LD;
    for (int e = lane; e < D * LD; e += kWave) Zt[e] = ((e / LD) == (e % LD)) ? 1.0 : 0.0;
    wave_lds_fence();
    double fro2 = 0.0;
    for (int e = lane; e < D * D; e += kWave) {
      const int i = e / D, j = e - i * D;
      fro2 += G[(i * LD + j) * 2] * G[(i * LD + j) * 2] + G[(i * LD + j) * 2 + 1] * G[(i * LD + j) * 2 + 1];
    }
    fro2 = wave_sum(fro2);
    double lam, xre[LD], xim[LD];
    if (wave_heev_ql<LD>(G, LD, Zt, LD, dv, ev, tauv, vbuf, wbuf, D, lane,
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
          vtop[r * 2] = xre[r];
          vtop[r * 2 + 1] = xim[r];
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
      const double ar = X[(m * LD + i) * 2], ai = X[(m * LD + i) * 2 + 1];
      const double ur = vtop[m * 2], ui = vtop[m * 2 + 1];
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
  hipLaunchKernelGGL(gen_ban_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, s, w, nn, N, D,
                     out);
  return ok_or_hip();
}

int launch_gen_gev(const double* t, const double* nn, int64_t N, int D, double* w, int32_t* st,
                   size_t lds_limit, hipStream_t s) {
  if (D < 2 || D > LD) return PBBSS_ERR_UNSUPPORTED;
  if (kGevBytes > lds_limit) return PBBSS_ERR_LDS_CAPACITY;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(gen_gev_kernel),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGevBytes) != hipSuccess)
    return PBBSS_ERR_HIP;
  hipLaunchKernelGGL(gen_gev_kernel, dim3((unsigned)N), dim3(kGenThreads), kGevBytes, s, t, nn, D,
                     w, st);
  return ok_or_hip();
}

}  // namespace pbbss
