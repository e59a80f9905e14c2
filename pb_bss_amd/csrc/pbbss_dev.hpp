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


// broadcast the value held by a COMPILE-TIME-known lane to the whole wave
// through SGPRs (v_readlane_b32): no LDS-crossbar round trip.
__device__ __forceinline__ double lane_bcast_const(double v, int src_lane_uniform) {
  union {
    double d;
    int i[2];
  } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], src_lane_uniform);
  u.i[1] = __builtin_amdgcn_readlane(u.i[1], src_lane_uniform);
  return u.d;
}

// Sum N per-lane values over the 64 lanes with a halving ("reduce-scatter")
// butterfly: the four steps over lane bits 5..2 each exchange only HALF of the
// remaining values (lanes with the bit clear keep the lower half, the others
// the upper half), two full steps over bits 1..0 finish.  N must be a multiple
// of 16.  On return v[0 .. N/16) hold the totals of the original indices
//     base + m,  base = (N/16) * (b2 + 2*b3 + 4*b4 + 8*b5)   (b_i = lane bit i)
// on every lane (lanes differing only in bits 1..0 hold identical values).
// All exchanges of a step are independent, so their ds_bpermute latencies overlap
// (the naive "one all-reduce per value" chain is ~10x slower: 6 dependent hops each).
template <int N>
__device__ __forceinline__ void wave_reduce_scatter(double (&v)[N], int lane) {
  static_assert(N % 16 == 0, "pad to a multiple of 16");
  static_for<0, 4>([&](auto sc) {
    constexpr int s = sc;
    constexpr int H = N >> (s + 1);
    constexpr int bit = 32 >> s;
    const bool up = (lane & bit) != 0;
    double send[H];
#pragma unroll
    for (int n = 0; n < H; ++n) {
      double lo = v[n], hi = v[n + H];
      send[n] = up ? lo : hi;
      v[n] = up ? hi : lo;
    }
#pragma unroll
    for (int n = 0; n < H; ++n) send[n] = __shfl_xor(send[n], bit, kWave);
#pragma unroll
    for (int n = 0; n < H; ++n) v[n] += send[n];
  });
  constexpr int R = N / 16;
#pragma unroll
  for (int o = 2; o > 0; o >>= 1) {
    double t[R];
#pragma unroll
    for (int n = 0; n < R; ++n) t[n] = __shfl_xor(v[n], o, kWave);
#pragma unroll
    for (int n = 0; n < R; ++n) v[n] += t[n];
  }
}
// first original index owned by `lane` after wave_reduce_scatter<N>
template <int N>
__device__ __forceinline__ int reduce_scatter_base(int lane) {
  return (N / 16) * ((lane >> 2) & 15);
}

// two interleaved all-reduce sums (halves the dependent-hop latency of two
// back-to-back wave_sum calls)
__device__ __forceinline__ void wave_sum2(double& a, double& b) {
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) {
    double ta = __shfl_xor(a, o, kWave), tb = __shfl_xor(b, o, kWave);
    a += ta;
    b += tb;
  }
}


// 1/x to ~1 ulp for finite normal x: hardware estimate + two Newton steps
// (5 VALU instructions instead of the ~11 of an IEEE-correct division).
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = fma(-x, r, 1.0);
  r = fma(e, r, r);
  e = fma(-x, r, 1.0);
  r = fma(e, r, r);
  return r;
}

constexpr double kTiny = 2.2250738585072014e-308;  // np.finfo(np.float64).tiny

}  // namespace pbbss
