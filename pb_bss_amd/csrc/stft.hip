// STFT / inverse STFT edge of the separation path: audio in,
// audio out without leaving the device.  The reference has no transform of its own; its
// tests call nara_wpe.utils.stft / istft (tests/test_distribution/test_spatial_mm.py:4,17-22),
// whose algorithm these kernels follow: fading pads, periodic analysis window, a frame every
// `shift` samples, rfft(n = size); the inverse multiplies irfft frames by the biorthogonal
// synthesis window and overlap-adds.
//
// One 256-thread workgroup transforms kFramesPerWg consecutive frames of one channel:
//   * a real frame of N = size samples is packed into M = N/2 complex points
//     z[n] = x[2n] + i x[2n+1], transformed by a radix-2 Stockham autosort FFT in LDS
//     (ping-pong buffers, one barrier per stage, no bit reversal), and untangled into the
//     N/2+1 bins of the real transform; the inverse runs the same steps backwards;
//   * one table of N-th roots of unity per workgroup (sincospi once per entry) serves both the
//     FFT stages (M-th roots = every second entry) and the untangling step;
//   * the forward kernel can write (channel, frame, bin) like the reference, or directly
//     (bin, frame, channel) - the 'd t f -> f t d' rearrangement every caller applies before
//     the mixture-model trainers (test_spatial_mm.py:41) - so the EM kernels read it as is.
// Float64 arithmetic (numpy's rfft on float64 input); complex64 or complex128 output.
// HBM traffic is tiny next to the EM loop (one utterance: 3.1 MB out), so the kernels are
// written for simplicity: the time is launch + LDS-latency bound, a few tens of microseconds.
#include "stft.hpp"
#include <cmath>
#include "pbbss_dev.hpp"

namespace pbbss {
namespace {

constexpr int kStftThreads = 256;
constexpr int kFramesPerWg = 4;

struct Cplx {
  double re, im;
};

__device__ __forceinline__ Cplx cmul(Cplx a, Cplx b) {
  return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}

// roots[k] = exp(-2 pi i k / N), k = 0..M (M = N/2)
__device__ void build_roots(Cplx* roots, int N, int tid) {
  const int M = N / 2;
  for (int k = tid; k <= M; k += kStftThreads) {
    double s, c;
    sincospi(-2.0 * (double)k / (double)N, &s, &c);
    roots[k] = {c, s};
  }
}

// In-LDS radix-2 Stockham FFT of M points (M a power of two >= 2).  SIGN = -1 forward,
// +1 backward (unscaled).  Returns the buffer that holds the result.
template <int SIGN>
__device__ Cplx* stockham(Cplx* a, Cplx* b, const Cplx* roots, int M, int tid) {
  int n = M, s = 1, ls = 0;  // s = 2^ls butterflies share a twiddle
  while (n > 1) {
    const int half = n >> 1;
    for (int j = tid; j < (M >> 1); j += kStftThreads) {
      const int p = j >> ls, q = j & (s - 1);
      Cplx w = roots[2 * (p << ls)];  // exp(-2 pi i p / n) = roots_N[2 p M / n]
      if (SIGN > 0) w.im = -w.im;
      const Cplx u = a[q + s * p], v = a[q + s * (p + half)];
      b[q + s * (2 * p)] = {u.re + v.re, u.im + v.im};
      b[q + s * (2 * p + 1)] = cmul({u.re - v.re, u.im - v.im}, w);
    }
    __syncthreads();
    Cplx* t = a;
    a = b;
    b = t;
    n = half;
    s <<= 1;
    ++ls;
  }
  return a;
}

struct StftArgs {
  const void* x;        // (C, N) float32 / float64
  int x_is_f64;
  int64_t C, N;
  int size, shift, wl;
  const double* window; // (wl) analysis window
  int fade;             // zeros in front of the signal (window_length - shift, or 0)
  int T;
  int layout;           // 0: out (C, T, F), 1: out (F, T, C)
  int out_c128;
  void* out;
};

__global__ void __launch_bounds__(kStftThreads) stft_kernel(StftArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int N = g.size, M = N / 2, F = M + 1;
  Cplx* bufa = reinterpret_cast<Cplx*>(smem);
  Cplx* bufb = bufa + M;
  Cplx* roots = bufb + M;  // [M + 1]
  const int tid = threadIdx.x;
  const int64_t c = blockIdx.y;
  build_roots(roots, N, tid);
  for (int ft = 0; ft < kFramesPerWg; ++ft) {
    const int t = blockIdx.x * kFramesPerWg + ft;
    if (t >= g.T) break;  // uniform
    __syncthreads();      // roots ready / previous frame fully written out
    const int64_t s0 = (int64_t)t * g.shift - g.fade;
    for (int n = tid; n < M; n += kStftThreads) {
      // unconditional loads at clamped indices, masks afterwards (a guarded load that is used
      // inside its guard waits for its own memory round trip)
      bool ok[2];
      int64_t sc[2];
      int wc[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int m = 2 * n + h;
        const int64_t si = s0 + m;
        ok[h] = m < g.wl && si >= 0 && si < g.N;
        sc[h] = ok[h] ? si : 0;
        wc[h] = ok[h] ? m : 0;
      }
      double xv[2], wv[2];
      if (g.x_is_f64) {
        const double* xp = static_cast<const double*>(g.x) + c * g.N;
        xv[0] = xp[sc[0]];
        xv[1] = xp[sc[1]];
      } else {
        const float* xp = static_cast<const float*>(g.x) + c * g.N;
        const float f0 = xp[sc[0]], f1 = xp[sc[1]];
        xv[0] = (double)f0;
        xv[1] = (double)f1;
      }
      wv[0] = g.window[wc[0]];
      wv[1] = g.window[wc[1]];
      bufa[n] = {ok[0] ? xv[0] * wv[0] : 0.0, ok[1] ? xv[1] * wv[1] : 0.0};
    }
    __syncthreads();
    const Cplx* Z = stockham<-1>(bufa, bufb, roots, M, tid);
    // untangle: X[k] = Xe + exp(-2 pi i k / N) Xo with the even / odd sample spectra
    // Xe = (Z[k] + conj Z[M-k]) / 2, Xo = (Z[k] - conj Z[M-k]) / (2i)
    for (int k = tid; k < F; k += kStftThreads) {
      const Cplx zk = Z[k & (M - 1)], zc = Z[(M - k) & (M - 1)];
      const Cplx xe = {0.5 * (zk.re + zc.re), 0.5 * (zk.im - zc.im)};
      const Cplx xo = {0.5 * (zk.im + zc.im), -0.5 * (zk.re - zc.re)};
      const Cplx r = cmul(roots[k], xo);
      const double ore = xe.re + r.re, oim = xe.im + r.im;
      const size_t idx = (g.layout == 0) ? ((size_t)c * g.T + t) * F + k
                                         : ((size_t)k * g.T + t) * g.C + c;
      if (g.out_c128) {
        double* o = static_cast<double*>(g.out) + 2 * idx;
        o[0] = ore;
        o[1] = oim;
      } else {
        float* o = static_cast<float*>(g.out) + 2 * idx;
        o[0] = (float)ore;
        o[1] = (float)oim;
      }
    }
  }
}

struct IstftArgs {
  const void* X;          // (C, T, F) complex64 / complex128
  int x_is_c128;
  int64_t C;
  int T, size, shift, wl;
  const double* window;   // (wl) synthesis (biorthogonal) window
  double* frames;         // (C, T, wl) scratch: windowed inverse transforms
};

__global__ void __launch_bounds__(kStftThreads) istft_frames_kernel(IstftArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int N = g.size, M = N / 2, F = M + 1;
  Cplx* bufa = reinterpret_cast<Cplx*>(smem);
  Cplx* bufb = bufa + M;
  Cplx* roots = bufb + M;
  Cplx* spec = roots + F;  // [F] the frame's spectrum
  const int tid = threadIdx.x;
  const int64_t c = blockIdx.y;
  build_roots(roots, N, tid);
  for (int ft = 0; ft < kFramesPerWg; ++ft) {
    const int t = blockIdx.x * kFramesPerWg + ft;
    if (t >= g.T) break;
    __syncthreads();
    const size_t base = ((size_t)c * g.T + t) * F;
    for (int k = tid; k < F; k += kStftThreads) {
      double re, im;
      if (g.x_is_c128) {
        const double* p = static_cast<const double*>(g.X) + 2 * (base + k);
        re = p[0];
        im = p[1];
      } else {
        const float* p = static_cast<const float*>(g.X) + 2 * (base + k);
        re = (double)p[0];
        im = (double)p[1];
      }
      if (k == 0 || k == M) im = 0.0;  // numpy.fft.irfft ignores them
      spec[k] = {re, im};
    }
    __syncthreads();
    // Z[k] = Xe + i Xo, Xe = (X[k] + conj X[M-k]) / 2, Xo = (X[k] - conj X[M-k]) / 2 * exp(+2 pi i k / N)
    for (int k = tid; k < M; k += kStftThreads) {
      const Cplx xk = spec[k], xc = spec[M - k];
      const Cplx xe = {0.5 * (xk.re + xc.re), 0.5 * (xk.im - xc.im)};
      const Cplx d = {0.5 * (xk.re - xc.re), 0.5 * (xk.im + xc.im)};
      const Cplx w = {roots[k].re, -roots[k].im};
      const Cplx xo = cmul(d, w);
      bufa[k] = {xe.re - xo.im, xe.im + xo.re};
    }
    __syncthreads();
    const Cplx* z = stockham<+1>(bufa, bufb, roots, M, tid);
    const double scale = 1.0 / (double)M;
    double* o = g.frames + ((size_t)c * g.T + t) * g.wl;
    for (int n = tid; n < M; n += kStftThreads) {
      const int m = 2 * n;
      if (m < g.wl) o[m] = z[n].re * scale * g.window[m];
      if (m + 1 < g.wl) o[m + 1] = z[n].im * scale * g.window[m + 1];
    }
  }
}

struct OlaArgs {
  const double* frames;  // (C, T, wl)
  int64_t C;
  int T, shift, wl, fade;
  int64_t n_out;
  double* out;           // (C, n_out)
};

// thread = output sample: the frames covering it, added in ascending frame order
__global__ void __launch_bounds__(kStftThreads) istft_ola_kernel(OlaArgs g) {
  const int64_t n = (int64_t)blockIdx.x * kStftThreads + threadIdx.x;
  const int64_t c = blockIdx.y;
  if (n >= g.n_out) return;
  const int64_t m = n + g.fade;
  int64_t t_hi = m / g.shift;
  if (t_hi > g.T - 1) t_hi = g.T - 1;
  int64_t t_lo = (m - g.wl + g.shift) / g.shift;  // ceil((m - wl + 1) / shift) for m - wl + 1 > 0
  if (m - g.wl + 1 <= 0) t_lo = 0;
  double acc = 0.0;
  for (int64_t t = t_lo; t <= t_hi; ++t) acc += g.frames[((size_t)c * g.T + t) * g.wl + (m - t * g.shift)];
  g.out[c * g.n_out + n] = acc;
}

bool pow2(int v) { return v >= 4 && (v & (v - 1)) == 0; }

}  // namespace

int launch_stft(const void* x, int x_is_f64, int64_t C, int64_t N, int size, int shift, int wl,
                const double* window, int fade, int T, int layout, int out_c128, void* out,
                size_t lds_limit, hipStream_t s) {
  if (!pow2(size) || size > 8192 || wl < 1 || wl > size || shift < 1 || T < 1 || C < 1 ||
      C > 65535 || layout < 0 || layout > 1)
    return PBBSS_ERR_UNSUPPORTED;
  const int M = size / 2;
  const size_t lds = ((size_t)2 * M + M + 1) * sizeof(Cplx);
  if (lds > lds_limit) return PBBSS_ERR_LDS_CAPACITY;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(stft_kernel),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return PBBSS_ERR_HIP;
  StftArgs g{x, x_is_f64, C, N, size, shift, wl, window, fade, T, layout, out_c128, out};
  const dim3 grid((unsigned)((T + kFramesPerWg - 1) / kFramesPerWg), (unsigned)C);
  hipLaunchKernelGGL(stft_kernel, grid, dim3(kStftThreads), lds, s, g);
  return hipGetLastError() == hipSuccess ? PBBSS_OK : PBBSS_ERR_HIP;
}

int launch_istft(const void* X, int x_is_c128, int64_t C, int T, int size, int shift, int wl,
                 const double* window, int fade, double* frames, double* out, int64_t n_out,
                 size_t lds_limit, hipStream_t s) {
  if (!pow2(size) || size > 8192 || wl < 1 || wl > size || shift < 1 || T < 1 || C < 1 ||
      C > 65535 || n_out < 1)
    return PBBSS_ERR_UNSUPPORTED;
  const int M = size / 2;
  const size_t lds = ((size_t)2 * M + 2 * (M + 1)) * sizeof(Cplx);
  if (lds > lds_limit) return PBBSS_ERR_LDS_CAPACITY;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(istft_frames_kernel),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return PBBSS_ERR_HIP;
  IstftArgs g{X, x_is_c128, C, T, size, shift, wl, window, frames};
  const dim3 grid((unsigned)((T + kFramesPerWg - 1) / kFramesPerWg), (unsigned)C);
  hipLaunchKernelGGL(istft_frames_kernel, grid, dim3(kStftThreads), lds, s, g);
  OlaArgs o{frames, C, T, shift, wl, fade, n_out, out};
  const dim3 grid2((unsigned)((n_out + kStftThreads - 1) / kStftThreads), (unsigned)C);
  hipLaunchKernelGGL(istft_ola_kernel, grid2, dim3(kStftThreads), 0, s, o);
  return hipGetLastError() == hipSuccess ? PBBSS_OK : PBBSS_ERR_HIP;
}

}  // namespace pbbss
