// Device-side helpers shared by the gfx950 kernels: wave64 cross-lane
// primitives and compile-time loops.  CDNA4 only (64-lane wavefronts).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <utility>

namespace pbbss {

constexpr int kWave = 64;       // CDNA wavefront width
constexpr int kEmThreads = 256; // EM workgroup: one wave per SIMD of a CU
constexpr int kEmWaves = kEmThreads / kWave;

// ---- compile-time loop: body(std::integral_constant<int, I>{}) for I in [B, E)
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(static_cast<F&&>(f));
  }
}

// ---- strict-upper-triangle enumeration: p in [0, D(D-1)/2) <-> (i < j), row major
template <int D>
__host__ __device__ constexpr int tri_i(int p) {
  int i = 0;
  while (p >= D - 1 - i) {
    p -= D - 1 - i;
    ++i;
  }
  return i;
}
template <int D>
__host__ __device__ constexpr int tri_j(int p) {
  int i = 0;
  while (p >= D - 1 - i) {
    p -= D - 1 - i;
    ++i;
  }
  return i + 1 + p;
}

// ---- wave64 cross-lane --------------------------------------------------
__device__ __forceinline__ double lane_get(double v, int src_lane) {
  return __shfl(v, src_lane, kWave);
}
__device__ __forceinline__ int lane_get(int v, int src_lane) {
  return __shfl(v, src_lane, kWave);
}
// all-reduce sum over the 64 lanes (every lane receives the total)
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, kWave));
  return v;
}
__device__ __forceinline__ int wave_or(int v) {
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) v |= __shfl_xor(v, o, kWave);
  return v;
}

// x^N for a compile-time non-negative integer N by binary powering
template <int N>
__device__ __forceinline__ double ipow(double x) {
  if constexpr (N == 0) {
    return 1.0;
  } else if constexpr (N == 1) {
    return x;
  } else if constexpr (N % 2 == 0) {
    double h = ipow<N / 2>(x);
    return h * h;
  } else {
    return x * ipow<N - 1>(x);
  }
}

// mantissa/exponent pair representing m * 2^e; keeps products of many
// eigenvalues / pivots inside the double range.
struct ScaledReal {
  double m;
  int e;
};
__device__ __forceinline__ void scaled_mul(ScaledReal& s, double x) {
  int ex;
  double mx = frexp(x, &ex);
  s.m *= mx;
  s.e += ex;
  int e2;
  s.m = frexp(s.m, &e2);
  s.e += e2;
}

constexpr double kTiny = 2.2250738585072014e-308;  // np.finfo(np.float64).tiny

}  // namespace pbbss
