// Mask-based beamformer extraction kernels for gfx950: PSD estimation and the
// batched small dense solvers behind GEV / MVDR / BAN.  One wavefront per
// D x D problem (lane = matrix entry, see wave_la.hpp), four problems per
// 256-thread workgroup; the streaming ops (normalize, apply) are lane = frame.
//
// Reference functions replaced (paths under /root/reference/pb_bss/):
//   extraction/beamformer.py:59-160   get_power_spectral_density_matrix
//   extraction/beamformer.py:292-364  get_gev_vector (-> cythonized zhegvd loop)
//   extraction/beamformer.py:627-698  get_mvdr_vector_souden
//   extraction/beamformer.py:230-260  get_mvdr_vector
//   extraction/beamformer.py:459-488  blind_analytic_normalization
//   extraction/beamformer.py:572-583  apply_beamforming_vector
//   math/solve.py:20-114              stable_solve (the np.linalg.solve part)
//   distribution/complex_angular_central_gaussian.py:34-55 normalize_observation
#include "beamform.hpp"
#include "cacgmm_em.hpp"

namespace pbbss {

constexpr int kLaThreads = 256;
constexpr int kLaWaves = kLaThreads / kWave;

// ------------------------------------------------------------------ helpers
template <int D>
__device__ __forceinline__ void load_mat(const double* base, int64_t n, LaneIJ c, double& re,
                                         double& im) {
  re = 0.0;
  im = 0.0;
  if (c.i < D && c.j < D) {
    const double* p = base + ((n * D + c.i) * D + c.j) * 2;
    re = p[0];
    im = p[1];
  }
}
// Hermitian matrix defined by the UPPER triangle of the stored array (what
// LAPACK reads through the Cython wrapper: UPLO='L' of the Fortran-order
// transpose, cythonized/get_gev_vector.pyx:72-74, beamformer.py:324-331).
template <int D>
__device__ __forceinline__ void load_herm_upper(const double* base, int64_t n, LaneIJ c,
                                                double& re, double& im) {
  re = 0.0;
  im = 0.0;
  if (c.i < D && c.j < D) {
    int i = min(c.i, c.j), j = max(c.i, c.j);
    const double* p = base + ((n * D + i) * D + j) * 2;
    re = p[0];
    im = (c.i == c.j) ? 0.0 : ((c.i < c.j) ? p[1] : -p[1]);
  }
}
template <int D>
__device__ __forceinline__ void store_mat(double* base, int64_t n, LaneIJ c, double re,
                                          double im) {
  if (c.i < D && c.j < D) {
    double* p = base + ((n * D + c.i) * D + c.j) * 2;
    p[0] = re;
    p[1] = im;
  }
}

// ------------------------------------------------------------------ heev
template <int D>
__global__ void __launch_bounds__(kLaThreads) heev_kernel(const double* a, int64_t N,
                                                          double* out_val, double* out_vec,
                                                          int32_t* status) {
  const int lane = threadIdx.x & 63;
  // per-round lane bookkeeping of the Jacobi, once per workgroup (wave_la.hpp)
  __shared__ uint32_t jtab[jacobi_table_dwords<D>()];
  if (threadIdx.x < kWave) jacobi_table_build<D>(jtab, lane);
  __syncthreads();
  const int64_t n = (int64_t)blockIdx.x * kLaWaves + (threadIdx.x >> 6);
  if (n >= N) return;  // wave-uniform
  const LaneIJ c = lane_ij(lane);
  double are = 0.0, aim = 0.0, vre, vim;
  // numpy.linalg.eigh(a) (UPLO='L') reads only the LOWER triangle of the
  // row-major array; mirror that choice exactly:
  if (c.i < D && c.j < D) {
    int i = max(c.i, c.j), j = min(c.i, c.j);
    const double* p = a + ((n * D + i) * D + j) * 2;
    are = p[0];
    aim = (c.i == c.j) ? 0.0 : ((c.i > c.j) ? p[1] : -p[1]);
  }
  int sweeps = wave_jacobi_heev_tab<D>(are, aim, c, vre, vim, jtab, lane);
  double lam = lane_get(are, ij_lane(c.j, c.j));
  int rank = wave_sort_rank<D>(lam, c);
  if (c.i < D && c.j < D) {
    double* ov = out_vec + ((n * D + c.i) * D + rank) * 2;
    ov[0] = vre;
    ov[1] = vim;
    if (c.i == 0) out_val[n * D + rank] = lam;
  }
  if (status && lane == 0) status[n] = (sweeps < 0) ? PBBSS_ST_EIG_NOCONV : 0;
}

// ------------------------------------------------------------------ GEV
// Phi_xx w = lambda Phi_nn w via Phi_nn = L L^H, M = L^-1 Phi_xx L^-H (Hermitian),
// M u = lambda u, w = L^-H u  (what zhegvd does); w^H Phi_nn w = 1.
template <int D>
__global__ void __launch_bounds__(kLaThreads) gev_kernel(const double* target,
                                                         const double* noise, int64_t N,
                                                         double* out_w, int32_t* status) {
  const int lane = threadIdx.x & 63;
  __shared__ uint32_t jtab[jacobi_table_dwords<D>()];
  if (threadIdx.x < kWave) jacobi_table_build<D>(jtab, lane);
  __syncthreads();
  const int64_t n = (int64_t)blockIdx.x * kLaWaves + (threadIdx.x >> 6);
  if (n >= N) return;
  const LaneIJ c = lane_ij(lane);
  double tre, tim, nre, nim;
  load_herm_upper<D>(target, n, c, tre, tim);
  load_herm_upper<D>(noise, n, c, nre, nim);
  int st = 0;
  bool fin = isfinite(tre) && isfinite(tim) && isfinite(nre) && isfinite(nim);
  if (wave_or(fin ? 0 : 1)) st |= PBBSS_ST_NONFINITE;
  ScaledReal det;
  int info = wave_cholesky<D>(nre, nim, c, det);
  if (info != 0) st |= PBBSS_ST_NOT_POSDEF | (info << 8);
  // keep only the lower triangle of L
  if (c.i < c.j) {
    nre = 0.0;
    nim = 0.0;
  }
  double xre, xim;
  wave_tri_inverse<D>(nre, nim, c, xre, xim);  // X = L^-1 (lower)
  if (c.i < c.j || c.i >= D || c.j >= D) {
    xre = 0.0;
    xim = 0.0;
  }
  double pre, pim, xhre, xhim, mre, mim;
  wave_matmul<D>(xre, xim, tre, tim, c, pre, pim);     // X Phi_xx
  wave_adjoint(xre, xim, c, xhre, xhim);               // X^H
  wave_matmul<D>(pre, pim, xhre, xhim, c, mre, mim);   // M = X Phi_xx X^H
  // symmetrise rounding noise
  double mtre, mtim;
  wave_adjoint(mre, mim, c, mtre, mtim);
  mre = 0.5 * (mre + mtre);
  mim = 0.5 * (mim + mtim);
  double vre, vim;
  int sweeps = wave_jacobi_heev_tab<D>(mre, mim, c, vre, vim, jtab, lane);
  if (sweeps < 0) st |= PBBSS_ST_EIG_NOCONV;
  double lam = lane_get(mre, ij_lane(c.j, c.j));
  int rank = wave_sort_rank<D>(lam, c);
  // the principal eigenvector sits in the column whose rank is D-1
  int col = 0;
#pragma unroll
  for (int m = 0; m < D; ++m) {
    int rm = lane_get(rank, ij_lane(0, m));
    if (rm == D - 1) col = m;
  }
  // w_i = sum_m conj(X_mi) u_m
  double wre = 0.0, wim = 0.0;
#pragma unroll
  for (int m = 0; m < D; ++m) {
    double ar = lane_get(xre, ij_lane(m, c.i)), ai = lane_get(xim, ij_lane(m, c.i));
    double ur = lane_get(vre, ij_lane(m, col)), ui = lane_get(vim, ij_lane(m, col));
    wre += ar * ur + ai * ui;
    wim += ar * ui - ai * ur;
  }
  if (c.j == 0 && c.i < D) {
    double* o = out_w + (n * D + c.i) * 2;
    o[0] = wre;
    o[1] = wim;
  }
  if (status && lane == 0) status[n] = st;
}

// ------------------------------------------------------------------ solve
template <int D>
__global__ void __launch_bounds__(kLaThreads) solve_kernel(const double* A, const double* Bm,
                                                           int64_t N, int M, double* out,
                                                           int32_t* status) {
  const int lane = threadIdx.x & 63;
  const int64_t n = (int64_t)blockIdx.x * kLaWaves + (threadIdx.x >> 6);
  if (n >= N) return;
  const LaneIJ c = lane_ij(lane);
  double are, aim, bre = 0.0, bim = 0.0, xre, xim;
  load_mat<D>(A, n, c, are, aim);
  if (c.i < D && c.j < M) {
    const double* p = Bm + ((n * D + c.i) * M + c.j) * 2;
    bre = p[0];
    bim = p[1];
  }
  bool sing = wave_lu_solve<D>(are, aim, bre, bim, c, xre, xim);
  if (sing) wave_pinv_solve<D>(are, aim, bre, bim, c, xre, xim);  // math/solve.py:111-113
  if (c.i < D && c.j < M) {
    double* p = out + ((n * D + c.i) * M + c.j) * 2;
    p[0] = xre;
    p[1] = xim;
  }
  if (status && lane == 0) status[n] = sing ? PBBSS_ST_SINGULAR : 0;
}

// ------------------------------------------------------------------ MVDR (Souden)
// mode 0: MVDR-Souden   mat = G / max(Re tr G, eps)                 (beamformer.py:683-686)
// mode 1: wMWF          mat = G / (mu + tr G)                       (beamformer.py:736-742)
// mode 2: wMWF 'frequency_dependent'  mat = G / sqrt(target_00 * tr G)   (:737-740)
template <int D>
__global__ void __launch_bounds__(kLaThreads)
    mvdr_souden_kernel(const double* target, const double* noise, int64_t N, double eps,
                       int mode, double* out_mat, double* snr_num, double* snr_den,
                       int32_t* status) {
  const int lane = threadIdx.x & 63;
  const int64_t n = (int64_t)blockIdx.x * kLaWaves + (threadIdx.x >> 6);
  if (n >= N) return;
  const LaneIJ c = lane_ij(lane);
  const bool valid = c.i < D && c.j < D;
  double tre, tim, nre, nim, gre, gim;
  load_mat<D>(target, n, c, tre, tim);
  load_mat<D>(noise, n, c, nre, nim);
  bool sing = wave_lu_solve<D>(nre, nim, tre, tim, c, gre, gim);  // G = noise^-1 target  (:682)
  if (sing) wave_pinv_solve<D>(nre, nim, tre, tim, c, gre, gim);  // stable_solve's lstsq branch
  // lambda = trace(G); mat = G / max(lambda.real, eps)            (:683-686)
  double tr = wave_sum((valid && c.i == c.j) ? gre : 0.0);
  if (mode == 0) {
    double sc = 1.0 / fmax(tr, eps);
    gre *= sc;
    gim *= sc;
  } else {
    // complex denominator: lambda = tr G is complex in the reference's wMWF
    double ti = wave_sum((valid && c.i == c.j) ? gim : 0.0);
    double dr, di;
    if (mode == 1) {
      dr = eps + tr;  // eps carries the distortion weight mu
      di = ti;
    } else {
      // sqrt(phi_x1x1 * lambda), principal branch
      double pr = lane_bcast_const(tre, 0), pi = lane_bcast_const(tim, 0);
      double zr = pr * tr - pi * ti, zi = pr * ti + pi * tr;
      double mag = sqrt(sqrt(zr * zr + zi * zi));
      double ang = 0.5 * atan2(zi, zr);
      dr = mag * cos(ang);
      di = mag * sin(ang);
    }
    double den = dr * dr + di * di;
    double nr = (gre * dr + gim * di) / den, ni = (gim * dr - gre * di) / den;
    gre = nr;
    gim = ni;
  }
  if (!valid) {
    gre = 0.0;
    gim = 0.0;
  }
  store_mat<D>(out_mat, n, c, gre, gim);
  // per-matrix terms of get_optimal_reference_channel (:616-620):
  //   num_r = sum_{d,e} conj(mat_dr) target_de mat_er,  den_r likewise with noise
  if (snr_num || snr_den) {
    double are, aim, bre, bim;
    wave_matmul<D>(tre, tim, gre, gim, c, are, aim);  // (target mat)_{d r} on lane (d, r)
    wave_matmul<D>(nre, nim, gre, gim, c, bre, bim);
    // column-wise inner product with conj(mat): reduce over rows i for each column j
    double pr = gre * are + gim * aim, pi = gre * aim - gim * are;  // conj(mat) * (target mat)
    double qr = gre * bre + gim * bim, qi = gre * bim - gim * bre;
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {  // sum over the row index (lane bits 3..5)
      pr += __shfl_xor(pr, o, kWave);
      pi += __shfl_xor(pi, o, kWave);
      qr += __shfl_xor(qr, o, kWave);
      qi += __shfl_xor(qi, o, kWave);
    }
    if (c.i == 0 && c.j < D) {
      if (snr_num) {
        snr_num[(n * D + c.j) * 2] = pr;
        snr_num[(n * D + c.j) * 2 + 1] = pi;
      }
      if (snr_den) {
        snr_den[(n * D + c.j) * 2] = qr;
        snr_den[(n * D + c.j) * 2 + 1] = qi;
      }
    }
  }
  if (status && lane == 0) status[n] = sing ? PBBSS_ST_SINGULAR : 0;
}

// ------------------------------------------------------------------ MVDR (ATF form)
template <int D>
__global__ void __launch_bounds__(kLaThreads) mvdr_kernel(const double* atf, const double* noise,
                                                          int64_t N, double* out_w,
                                                          int32_t* status) {
  const int lane = threadIdx.x & 63;
  const int64_t n = (int64_t)blockIdx.x * kLaWaves + (threadIdx.x >> 6);
  if (n >= N) return;
  const LaneIJ c = lane_ij(lane);
  double nre, nim, tre, tim;
  load_mat<D>(noise, n, c, nre, nim);
  wave_adjoint(nre, nim, c, tre, tim);  // hermitise (:246-248)
  nre = 0.5 * (nre + tre);
  nim = 0.5 * (nim + tim);
  double hre = 0.0, him = 0.0;
  if (c.j == 0 && c.i < D) {
    hre = atf[(n * D + c.i) * 2];
    him = atf[(n * D + c.i) * 2 + 1];
  }
  double xre, xim;
  bool sing = wave_lu_solve<D>(nre, nim, hre, him, c, xre, xim);  // numerator (:250)
  if (sing) wave_pinv_solve<D>(nre, nim, hre, him, c, xre, xim);  // lstsq fallback (:251-256)
  // denominator = h^H x (:257)
  double dr = (c.j == 0 && c.i < D) ? (hre * xre + him * xim) : 0.0;
  double di = (c.j == 0 && c.i < D) ? (hre * xim - him * xre) : 0.0;
  dr = wave_sum(dr);
  di = wave_sum(di);
  double den = dr * dr + di * di;
  // x / (dr + i di)
  double wr = (xre * dr + xim * di) / den, wi = (xim * dr - xre * di) / den;
  if (c.j == 0 && c.i < D) {
    out_w[(n * D + c.i) * 2] = wr;
    out_w[(n * D + c.i) * 2 + 1] = wi;
  }
  if (status && lane == 0) status[n] = sing ? PBBSS_ST_SINGULAR : 0;
}

// ------------------------------------------------------------------ BAN
template <int D>
__global__ void __launch_bounds__(kLaThreads) ban_kernel(const double* w, const double* noise,
                                                         int64_t N, double* out_w) {
  const int lane = threadIdx.x & 63;
  const int64_t n = (int64_t)blockIdx.x * kLaWaves + (threadIdx.x >> 6);
  if (n >= N) return;
  const LaneIJ c = lane_ij(lane);
  const bool valid = c.i < D && c.j < D;
  double nre, nim;
  load_mat<D>(noise, n, c, nre, nim);
  // w_i on every lane of row i; w_j on every lane of column j
  double wir = 0, wii = 0, wjr = 0, wji = 0;
  if (c.i < D) {
    wir = w[(n * D + c.i) * 2];
    wii = w[(n * D + c.i) * 2 + 1];
  }
  if (c.j < D) {
    wjr = w[(n * D + c.j) * 2];
    wji = w[(n * D + c.j) * 2 + 1];
  }
  // u_i = sum_j Phi_ij w_j (row sums), v_j = sum_i conj(w_i) Phi_ij (column sums)
  double ur = valid ? (nre * wjr - nim * wji) : 0.0, ui = valid ? (nre * wji + nim * wjr) : 0.0;
  double vr = valid ? (wir * nre + wii * nim) : 0.0, vi = valid ? (wir * nim - wii * nre) : 0.0;
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) {  // over j (lane bits 0..2)
    ur += __shfl_xor(ur, o, kWave);
    ui += __shfl_xor(ui, o, kWave);
  }
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) {  // over i (lane bits 3..5)
    vr += __shfl_xor(vr, o, kWave);
    vi += __shfl_xor(vi, o, kWave);
  }
  // u_i lives on all lanes of row i, v_j on all lanes of column j.
  // nominator  = sum_b v_b u_b  (:473-476);  denominator = sum_a conj(w_a) u_a (:479-481)
  double ubr = lane_get(ur, ij_lane(c.j, 0)), ubi = lane_get(ui, ij_lane(c.j, 0));  // u_j
  double nr = (c.i == 0 && c.j < D) ? (vr * ubr - vi * ubi) : 0.0;
  double ni = (c.i == 0 && c.j < D) ? (vr * ubi + vi * ubr) : 0.0;
  double dr = (c.i == 0 && c.j < D) ? (wjr * ubr + wji * ubi) : 0.0;
  double di = (c.i == 0 && c.j < D) ? (wjr * ubi - wji * ubr) : 0.0;
  nr = wave_sum(nr);
  ni = wave_sum(ni);
  dr = wave_sum(dr);
  di = wave_sum(di);
  double dabs = sqrt(dr * dr + di * di);
  double scale = (dabs != 0.0) ? sqrt(sqrt(nr * nr + ni * ni)) / dabs : 0.0;  // |sqrt(nom)/|den||
  if (c.j == 0 && c.i < D) {
    out_w[(n * D + c.i) * 2] = wir * scale;
    out_w[(n * D + c.i) * 2 + 1] = wii * scale;
  }
}

// ------------------------------------------------------------------ apply
// out[b,t] = sum_d conj(w[b,d]) x[b,d,t]; lane = frame, coalesced along t.
// xmod > 0: the observation has xmod problems, shared by B = m * xmod vectors (problem b reads
// x[b % xmod] -- K beamformers per bin on one STFT without K copies of it)
template <typename YS>
__global__ void __launch_bounds__(256) apply_kernel(const double* w, const void* xv, int64_t B,
                                                    int T, int D, double* out, int64_t xmod,
                                                    int64_t b_first) {
  using YS2 = typename std::conditional<std::is_same<YS, float>::value, float2, double2>::type;
  const YS2* x = reinterpret_cast<const YS2*>(xv);
  const int64_t b = blockIdx.y;
  const int64_t bx = xmod > 0 ? (b_first + b) % xmod : b;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < T; t += gridDim.x * blockDim.x) {
    double ar = 0.0, ai = 0.0;
    for (int d = 0; d < D; ++d) {
      double wr = w[(b * D + d) * 2], wi = w[(b * D + d) * 2 + 1];
      YS2 v = x[((size_t)bx * D + d) * T + t];
      double xr = (double)v.x, xi = (double)v.y;
      ar += wr * xr + wi * xi;
      ai += wr * xi - wi * xr;
    }
    out[((size_t)b * T + t) * 2] = ar;
    out[((size_t)b * T + t) * 2 + 1] = ai;
  }
}

// ------------------------------------------------------------------ normalize
// (B,T,D) -> unit norm over D -> (B,D,T).  A 64-frame tile goes through LDS so
// both the read (frame-major) and the write (sensor-major) are coalesced.
template <typename YS>
__global__ void __launch_bounds__(256) normalize_kernel(const void* yv, int64_t B, int T, int D,
                                                        void* outv) {
  using YS2 = typename std::conditional<std::is_same<YS, float>::value, float2, double2>::type;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  YS2* tile = reinterpret_cast<YS2*>(smem);  // [64][D+1]
  const YS2* y = reinterpret_cast<const YS2*>(yv);
  YS2* out = reinterpret_cast<YS2*>(outv);
  const int64_t b = blockIdx.y;
  const int t0 = blockIdx.x * 64;
  const int nt = min(64, T - t0);
  const int ld = D + 1;
  // coalesced read of nt*D contiguous elements
  for (int e = threadIdx.x; e < nt * D; e += blockDim.x) {
    int tt = e / D, d = e % D;
    tile[tt * ld + d] = y[((size_t)b * T + t0) * D + e];
  }
  __syncthreads();
  if (threadIdx.x < nt) {
    int tt = threadIdx.x;
    double n2 = 0.0;
    for (int d = 0; d < D; ++d) {
      YS2 v = tile[tt * ld + d];
      n2 += (double)v.x * (double)v.x + (double)v.y * (double)v.y;
    }
    // eps_style='where': zero norm -> divide by tiny -> stays 0 (utils.py:251)
    double inv = (n2 > 0.0) ? 1.0 / sqrt(n2) : 0.0;
    for (int d = 0; d < D; ++d) {
      YS2 v = tile[tt * ld + d];
      v.x = (YS)((double)v.x * inv);
      v.y = (YS)((double)v.y * inv);
      tile[tt * ld + d] = v;
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < nt * D; e += blockDim.x) {
    int d = e / nt, tt = e % nt;
    out[((size_t)b * D + d) * T + t0 + tt] = tile[tt * ld + d];
  }
}

// ------------------------------------------------------------------ PSD
// Reuses the EM kernel's LDS staging and its entry-split accumulation phase
// (phase M) with w_kt = normalised mask.
template <int D, int K, typename YS, bool SPILL>
__global__ void __launch_bounds__(kEmThreads, 3)
    psd_kernel(const void* x, int64_t B, int T, const double* mask, int64_t mask_bstride,
               int normalize, double* out, int64_t out_bstride, char* scratch,
               size_t scratch_stride) {
  using Kern = EmKernel<D, K, YS, SPILL>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  EmArgs a{};
  a.y = x;
  a.B = B;
  a.T = T;
  a.layout = PBBSS_LAYOUT_DT;
  const typename Kern::Lds L =
      Kern::carve(smem, T, SPILL ? scratch + (size_t)blockIdx.x * scratch_stride : nullptr);
  for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();
    if (tid == 0) *L.flags = 0;
    __syncthreads();
    Kern::phase_load(a, L, b, tid);
    // mask sums over frames (beamformer.py:127-131)
    double s[K];
#pragma unroll
    for (int k = 0; k < K; ++k) s[k] = 0.0;
    if (mask) {
      for (int t = tid; t < T; t += kEmThreads) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
          double m = mask[b * mask_bstride + (int64_t)k * T + t];
          L.wbuf[Kern::woff(k, t)] = m;
          s[k] += m;
        }
      }
    } else {
      for (int t = tid; t < T; t += kEmThreads) L.wbuf[Kern::woff(0, t)] = 1.0 / (double)T;  // :114-117
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      double tot = wave_sum(s[k]);
      if (lane == 0) L.red[wave * K + k] = tot;
    }
    __syncthreads();
    if (mask && normalize) {
      double inv[K];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < kEmWaves; ++w) tot += L.red[w * K + k];
        inv[k] = 1.0 / fmax(tot, 1e-10);
      }
      for (int t = tid; t < T; t += kEmThreads) {
#pragma unroll
        for (int k = 0; k < K; ++k) L.wbuf[Kern::woff(k, t)] *= inv[k];
      }
    }
    __syncthreads();
    switch (wave) {
      case 0: Kern::template phase_m<0>(a, L, lane); break;
      case 1: Kern::template phase_m<1>(a, L, lane); break;
      case 2: Kern::template phase_m<2>(a, L, lane); break;
      default: Kern::template phase_m<3>(a, L, lane); break;
    }
    __syncthreads();
    for (int e = tid; e < K * D * D; e += kEmThreads) {
      const int k = e / (D * D), ij = e % (D * D);
      double re, im;
      Kern::cov_entry(L, k, ij / D, ij % D, re, im);
      double* o = out + b * out_bstride + ((int64_t)k * D * D + ij) * 2;
      o[0] = re;
      o[1] = im;
    }
  }
}

// ------------------------------------------------------------------ host dispatch
#define PBBSS_DISPATCH_D(D_, ...)    \
  switch (D_) {                      \
    case 2: { constexpr int DD = 2; __VA_ARGS__; } break; \
    case 3: { constexpr int DD = 3; __VA_ARGS__; } break; \
    case 4: { constexpr int DD = 4; __VA_ARGS__; } break; \
    case 5: { constexpr int DD = 5; __VA_ARGS__; } break; \
    case 6: { constexpr int DD = 6; __VA_ARGS__; } break; \
    case 7: { constexpr int DD = 7; __VA_ARGS__; } break; \
    case 8: { constexpr int DD = 8; __VA_ARGS__; } break; \
    default: return PBBSS_ERR_UNSUPPORTED;         \
  }

static inline int check_launch() {
  return hipGetLastError() == hipSuccess ? PBBSS_OK : PBBSS_ERR_HIP;
}
static inline unsigned la_grid(int64_t N) { return (unsigned)((N + kLaWaves - 1) / kLaWaves); }

int launch_heev(const double* a, int64_t N, int D, double* val, double* vec, int32_t* st,
                hipStream_t s) {
  PBBSS_DISPATCH_D(D, hipLaunchKernelGGL(heev_kernel<DD>, dim3(la_grid(N)), dim3(kLaThreads), 0,
                                         s, a, N, val, vec, st));
  return check_launch();
}
int launch_gev(const double* t, const double* nn, int64_t N, int D, double* w, int32_t* st,
               hipStream_t s) {
  PBBSS_DISPATCH_D(D, hipLaunchKernelGGL(gev_kernel<DD>, dim3(la_grid(N)), dim3(kLaThreads), 0, s,
                                         t, nn, N, w, st));
  return check_launch();
}
int launch_solve(const double* A, const double* Bm, int64_t N, int D, int M, double* x,
                 int32_t* st, hipStream_t s) {
  PBBSS_DISPATCH_D(D, hipLaunchKernelGGL(solve_kernel<DD>, dim3(la_grid(N)), dim3(kLaThreads), 0,
                                         s, A, Bm, N, M, x, st));
  return check_launch();
}
int launch_mvdr_souden(const double* t, const double* nn, int64_t N, int D, double eps, int mode,
                       double* mat, double* num, double* den, int32_t* st, hipStream_t s) {
  PBBSS_DISPATCH_D(D, hipLaunchKernelGGL(mvdr_souden_kernel<DD>, dim3(la_grid(N)),
                                         dim3(kLaThreads), 0, s, t, nn, N, eps, mode, mat, num,
                                         den, st));
  return check_launch();
}
int launch_mvdr(const double* atf, const double* nn, int64_t N, int D, double* w, int32_t* st,
                hipStream_t s) {
  PBBSS_DISPATCH_D(D, hipLaunchKernelGGL(mvdr_kernel<DD>, dim3(la_grid(N)), dim3(kLaThreads), 0,
                                         s, atf, nn, N, w, st));
  return check_launch();
}
int launch_ban(const double* w, const double* nn, int64_t N, int D, double* out, hipStream_t s) {
  PBBSS_DISPATCH_D(D, hipLaunchKernelGGL(ban_kernel<DD>, dim3(la_grid(N)), dim3(kLaThreads), 0, s,
                                         w, nn, N, out));
  return check_launch();
}
int launch_apply(const double* w, const void* x, int x128, int64_t B, int T, int D, double* out,
                 hipStream_t s, int64_t xmod, int64_t b_first) {
  unsigned gx = (unsigned)((T + 255) / 256);
  if (gx > 64) gx = 64;
  dim3 grid(gx, (unsigned)B);
  if (x128)
    hipLaunchKernelGGL(apply_kernel<double>, grid, dim3(256), 0, s, w, x, B, T, D, out, xmod, b_first);
  else
    hipLaunchKernelGGL(apply_kernel<float>, grid, dim3(256), 0, s, w, x, B, T, D, out, xmod, b_first);
  return check_launch();
}

// ------------------------------------------------------------------ reference channel
// get_optimal_reference_channel (beamformer.py:601-624) + the column select of
// get_mvdr_vector_souden (:690-698) for L problems at once, after pbbss_mvdr_souden left
// mat (N,D,D) and the per-bin SNR terms num / den (N,D) behind: one workgroup per problem sums the
// terms over its F bins (thread = (slot, channel), slots in ascending order: bit-reproducible),
//   snr_r = sum_f num[f,r] / max(sum_f den[f,r], eps)        (np.maximum on complex numbers orders
//   by the real part first, then the imaginary part: the complex value is kept when it wins),
// takes the first arg-max of Re snr (np.argmax), copies column r of every bin's matrix to out_w and
// reports whether every SNR was finite (the reference's assert, :619).  Problem l, bin f is matrix
// n = l * lead_stride + f * bin_stride of the pbbss_mvdr_souden call.  It replaced ~25 elementwise
// launches of the host framework per call in the separation chain (profiles/r06_h_extraction_chain.txt).
constexpr int kRefThreads = 256;
__global__ void __launch_bounds__(kRefThreads)
    refchan_select_kernel(const double* __restrict__ mat, const double* __restrict__ num,
                          const double* __restrict__ den, int64_t F, int D, int64_t lead_stride,
                          int64_t bin_stride, double eps, double* __restrict__ out_w,
                          int32_t* __restrict__ out_ref, int32_t* __restrict__ out_ok) {
  __shared__ double red[4][kRefThreads];
  __shared__ double snr_re[32], snr_im[32];
  __shared__ int ref_s;
  const int64_t l = blockIdx.x;
  const int tid = threadIdx.x;
  const int S = kRefThreads / D;  // slots (D <= 32: S >= 8)
  const int slot = tid / D, r = tid - slot * D;
  double a[4] = {0.0, 0.0, 0.0, 0.0};
  if (slot < S) {
    for (int64_t f = slot; f < F; f += S) {
      const size_t n = (size_t)(l * lead_stride + f * bin_stride);
      a[0] += num[(n * D + r) * 2];
      a[1] += num[(n * D + r) * 2 + 1];
      a[2] += den[(n * D + r) * 2];
      a[3] += den[(n * D + r) * 2 + 1];
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) red[i][tid] = a[i];
  __syncthreads();
  if (tid < D) {
    double t[4] = {0.0, 0.0, 0.0, 0.0};
    for (int sl = 0; sl < S; ++sl) {
#pragma unroll
      for (int i = 0; i < 4; ++i) t[i] += red[i][sl * D + tid];
    }
    // max(den, eps) in NumPy's complex order, then the complex quotient
    double dr = t[2], di = t[3];
    const bool keep = (dr > eps) || (dr == eps && di >= 0.0);
    if (!keep) {
      dr = eps;
      di = 0.0;
    }
    // complex quotient by Smith's method (what NumPy's division does): no dr^2 + di^2, which
    // underflows to 0 for the floor eps = tiny and would turn 0 / tiny into NaN
    if (fabs(dr) >= fabs(di)) {
      const double rat = di / dr, dd = dr + di * rat;
      snr_re[tid] = (t[0] + t[1] * rat) / dd;
      snr_im[tid] = (t[1] - t[0] * rat) / dd;
    } else {
      const double rat = dr / di, dd = dr * rat + di;
      snr_re[tid] = (t[0] * rat + t[1]) / dd;
      snr_im[tid] = (t[1] * rat - t[0]) / dd;
    }
  }
  __syncthreads();
  if (tid == 0) {
    int best = 0;
    bool ok = true;
    for (int i = 0; i < D; ++i) {
      const double v = snr_re[i];
      ok = ok && (fabs(v) <= 1.79e308) && (fabs(snr_im[i]) <= 1.79e308);  // false for NaN / Inf
      // np.argmax: the first maximum; a NaN counts as the maximum
      if (i > 0 && !(snr_re[best] != snr_re[best]) && (v > snr_re[best] || v != v)) best = i;
    }
    ref_s = best;
    out_ref[l] = best;
    out_ok[l] = ok ? 1 : 0;
  }
  __syncthreads();
  const int ref = ref_s;
  for (int64_t e = tid; e < F * D; e += kRefThreads) {
    const int64_t f = e / D;
    const int i = (int)(e - f * D);
    const size_t n = (size_t)(l * lead_stride + f * bin_stride);
    const size_t src = ((n * D + i) * D + ref) * 2;
    out_w[((size_t)(l * F + f) * D + i) * 2] = mat[src];
    out_w[((size_t)(l * F + f) * D + i) * 2 + 1] = mat[src + 1];
  }
}

int launch_select_reference_channel(const double* mat, const double* num, const double* den,
                                    int64_t L, int64_t F, int D, int64_t lead_stride,
                                    int64_t bin_stride, double eps, double* out_w, int32_t* out_ref,
                                    int32_t* out_ok, hipStream_t s) {
  if (D < 1 || D > 32 || L < 1 || F < 1) return PBBSS_ERR_INVALID_ARG;
  hipLaunchKernelGGL(refchan_select_kernel, dim3((unsigned)L), dim3(kRefThreads), 0, s, mat, num,
                     den, F, D, lead_stride, bin_stride, eps, out_w, out_ref, out_ok);
  return check_launch();
}
int launch_normalize(const void* y, int is128, int64_t B, int T, int D, void* out,
                     hipStream_t s) {
  dim3 grid((unsigned)((T + 63) / 64), (unsigned)B);
  size_t lds = (size_t)64 * (D + 1) * (is128 ? 16 : 8);
  if (is128)
    hipLaunchKernelGGL(normalize_kernel<double>, grid, dim3(256), lds, s, y, B, T, D, out);
  else
    hipLaunchKernelGGL(normalize_kernel<float>, grid, dim3(256), lds, s, y, B, T, D, out);
  return check_launch();
}

template <int D, int K, typename YS, bool SPILL>
static int launch_psd_variant(const void* x, int64_t B, int T, const double* mask,
                              int64_t mask_bstride, int normalize, double* out,
                              int64_t out_bstride, const EmLaunchCfg& cfg, hipStream_t s) {
  using Kern = EmKernel<D, K, YS, SPILL>;
  size_t lds = Kern::lds_bytes(T);
  if (lds > cfg.lds_limit) return PBBSS_ERR_LDS_CAPACITY;
  auto kfn = psd_kernel<D, K, YS, SPILL>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return PBBSS_ERR_HIP;
  int occ = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kfn, kEmThreads, lds) != hipSuccess)
    return PBBSS_ERR_HIP;
  if (occ < 1) occ = 1;
  int64_t grid = (int64_t)cfg.num_cu * occ;
  if (grid > B) grid = B;
  char* scratch = nullptr;
  size_t stride = 0;
  if (SPILL) {
    stride = Kern::scratch_bytes(T);
    scratch = static_cast<char*>(cfg.get_scratch(cfg.scratch_ctx, stride * grid));
    if (!scratch) return PBBSS_ERR_HIP;
  }
  hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(kEmThreads), lds, s, x, B, T, mask,
                     mask_bstride, normalize, out, out_bstride, scratch, stride);
  return check_launch();
}

template <int D, int K, typename YS>
static int launch_psd_one(const void* x, int64_t B, int T, const double* mask,
                          int64_t mask_bstride, int normalize, double* out, int64_t out_bstride,
                          const EmLaunchCfg& cfg, hipStream_t s) {
  if (EmKernel<D, K, YS, false>::lds_bytes(T) <= cfg.lds_limit)
    return launch_psd_variant<D, K, YS, false>(x, B, T, mask, mask_bstride, normalize, out,
                                               out_bstride, cfg, s);
  return launch_psd_variant<D, K, YS, true>(x, B, T, mask, mask_bstride, normalize, out,
                                            out_bstride, cfg, s);
}

template <int D, typename YS>
static int launch_psd_k(int K, const void* x, int64_t B, int T, const double* mask,
                        int64_t mbs, int normalize, double* out, int64_t obs,
                        const EmLaunchCfg& cfg, hipStream_t s) {
  switch (K) {
    case 1: return launch_psd_one<D, 1, YS>(x, B, T, mask, mbs, normalize, out, obs, cfg, s);
    case 2: return launch_psd_one<D, 2, YS>(x, B, T, mask, mbs, normalize, out, obs, cfg, s);
    case 3: return launch_psd_one<D, 3, YS>(x, B, T, mask, mbs, normalize, out, obs, cfg, s);
    case 4: return launch_psd_one<D, 4, YS>(x, B, T, mask, mbs, normalize, out, obs, cfg, s);
    default: return PBBSS_ERR_UNSUPPORTED;
  }
}

int launch_psd(const void* x, int x128, int64_t B, int T, int D, int K, const double* mask,
               int normalize, double* out, const EmLaunchCfg& cfg, hipStream_t s) {
  // more than 4 sources: chunks of <= 4 classes per launch
  for (int k0 = 0; k0 < K; k0 += 4) {
    int kc = (K - k0 < 4) ? (K - k0) : 4;
    const double* m = mask ? mask + (int64_t)k0 * T : nullptr;
    double* o = out + (int64_t)k0 * D * D * 2;
    int rc = PBBSS_OK;
    PBBSS_DISPATCH_D(D, rc = x128 ? launch_psd_k<DD, double>(kc, x, B, T, m, (int64_t)K * T,
                                                             normalize, o, (int64_t)K * D * D * 2,
                                                             cfg, s)
                                  : launch_psd_k<DD, float>(kc, x, B, T, m, (int64_t)K * T,
                                                            normalize, o, (int64_t)K * D * D * 2,
                                                            cfg, s));
    if (rc != PBBSS_OK) return rc;
  }
  return PBBSS_OK;
}

}  // namespace pbbss
