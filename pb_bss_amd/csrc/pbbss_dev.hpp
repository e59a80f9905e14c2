// Device-side helpers shared by the gfx950 kernels: wave64 cross-lane
// primitives and compile-time loops.  CDNA4 only (64-lane wavefronts).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <utility>

// Device-side checks of the debug build (make debug: -DPBBSS_DEBUG_ASSERT): a failed check traps
// the wavefront, the launch then fails with hipErrorLaunchFailure instead of corrupting memory.
#ifdef PBBSS_DEBUG_ASSERT
#define PBBSS_DEV_ASSERT(cond)           \
  do {                                   \
    if (!(cond)) __builtin_trap();       \
  } while (0)
#else
#define PBBSS_DEV_ASSERT(cond) ((void)0)
#endif

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


// (i, j) of every strict-upper pair p as a constant table (run-time lookups; the
// constexpr functions above are for compile-time indices only)
template <int D>
struct TriTable {
  unsigned char i[D * (D - 1) / 2 + 1];
  unsigned char j[D * (D - 1) / 2 + 1];
  constexpr TriTable() : i{}, j{} {
    for (int p = 0; p < D * (D - 1) / 2; ++p) {
      i[p] = (unsigned char)tri_i<D>(p);
      j[p] = (unsigned char)tri_j<D>(p);
    }
  }
};
template <int D>
__device__ constexpr TriTable<D> kTriTable{};


// Hide a value's provenance from the optimiser.  Used on lane / thread ids at the top
// of each phase of the persistent EM loop: otherwise LICM hoists every lane-derived
// address computation out of the iteration loop, runs out of registers and spills
// them to scratch (~60 scratch loads per iteration); recomputing a few integer ops
// in place is cheaper.
__device__ __forceinline__ int opaque(int x) {
  asm volatile("" : "+v"(x));
  return x;
}

// ---- wave64 cross-lane --------------------------------------------------
__device__ __forceinline__ double lane_get(double v, int src_lane) {
  return __shfl(v, src_lane, kWave);
}
__device__ __forceinline__ int lane_get(int v, int src_lane) {
  return __shfl(v, src_lane, kWave);
}
// ---- DPP / permlane-swap building blocks (VALU only, no LDS crossbar) -------
// DPP controls (gfx9 encoding)
constexpr int kDppQuadXor1 = 0xB1;       // quad_perm [1,0,3,2]
constexpr int kDppQuadXor2 = 0x4E;       // quad_perm [2,3,0,1]
constexpr int kDppQuadIdent = 0xE4;      // quad_perm [0,1,2,3]
constexpr int kDppRowHalfMirror = 0x141; // lane l <-> 7-l inside each 8 lanes
constexpr int kDppRowMirror = 0x140;     // lane l <-> 15-l inside each 16-lane row
constexpr int kDppRowRor4 = 0x124, kDppRowRor8 = 0x128, kDppRowRor12 = 0x12C;

union F64Bits {
  double d;
  int i[2];
  unsigned u[2];
};

// result lane l = enabled(bank) ? src[perm(l)] : old[l]
template <int CTRL, int BANK>
__device__ __forceinline__ double dpp_f64(double old, double src) {
  F64Bits o, s, r;
  o.d = old;
  s.d = src;
  r.i[0] = __builtin_amdgcn_update_dpp(o.i[0], s.i[0], CTRL, 0xF, BANK, false);
  r.i[1] = __builtin_amdgcn_update_dpp(o.i[1], s.i[1], CTRL, 0xF, BANK, false);
  return r.d;
}
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int src) {
  return __builtin_amdgcn_update_dpp(src, src, CTRL, 0xF, 0xF, false);
}
// v_permlane32_swap: a[32..63] <-> b[0..31];  v_permlane16_swap: odd rows of a <-> even rows of b
__device__ __forceinline__ void swap32_f64(double& a, double& b) {
  F64Bits A, B;
  A.d = a;
  B.d = b;
  auto r0 = __builtin_amdgcn_permlane32_swap(A.u[0], B.u[0], false, false);
  auto r1 = __builtin_amdgcn_permlane32_swap(A.u[1], B.u[1], false, false);
  A.u[0] = r0[0];
  B.u[0] = r0[1];
  A.u[1] = r1[0];
  B.u[1] = r1[1];
  a = A.d;
  b = B.d;
}
__device__ __forceinline__ void swap16_f64(double& a, double& b) {
  F64Bits A, B;
  A.d = a;
  B.d = b;
  auto r0 = __builtin_amdgcn_permlane16_swap(A.u[0], B.u[0], false, false);
  auto r1 = __builtin_amdgcn_permlane16_swap(A.u[1], B.u[1], false, false);
  A.u[0] = r0[0];
  B.u[0] = r0[1];
  A.u[1] = r1[0];
  B.u[1] = r1[1];
  a = A.d;
  b = B.d;
}

// all-reduce over the 64 lanes with a binary op (every lane receives the result):
// quad permutes, 8- and 16-lane mirrors, then the two cross-row swaps.
template <typename Op>
__device__ __forceinline__ double wave_allreduce(double v, Op op) {
  v = op(v, dpp_f64<kDppQuadXor1, 0xF>(v, v));
  v = op(v, dpp_f64<kDppQuadXor2, 0xF>(v, v));
  v = op(v, dpp_f64<kDppRowHalfMirror, 0xF>(v, v));
  v = op(v, dpp_f64<kDppRowMirror, 0xF>(v, v));
  double a = v, b = v;
  swap16_f64(a, b);
  v = op(a, b);
  a = v;
  b = v;
  swap32_f64(a, b);
  return op(a, b);
}
__device__ __forceinline__ double wave_sum(double v) {
  return wave_allreduce(v, [](double x, double y) { return x + y; });
}
__device__ __forceinline__ double wave_max(double v) {
  return wave_allreduce(v, [](double x, double y) { return fmax(x, y); });
}
__device__ __forceinline__ int wave_or(int v) {
  v |= dpp_i32<kDppQuadXor1>(v);
  v |= dpp_i32<kDppQuadXor2>(v);
  v |= dpp_i32<kDppRowHalfMirror>(v);
  v |= dpp_i32<kDppRowMirror>(v);
  {
    auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    v = (int)(r[0] | r[1]);
  }
  {
    auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    v = (int)(r[0] | r[1]);
  }
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
// Bits 5 and 4 use v_permlane32_swap / v_permlane16_swap (gfx950): swapping the
// (lower, upper) value pair between partner lanes and adding leaves the lower
// value's pair-sum on the bit-clear lanes and the upper value's on the bit-set
// lanes -- two swaps and one add per output, no selects.  Bits 3 and 2 use DPP
// row rotations with bank masks, bits 1 and 0 DPP quad permutes.  Everything is
// VALU: no ds_bpermute, nothing on the LDS pipe.
template <int N>
__device__ __forceinline__ void wave_reduce_scatter(double (&v)[N], int lane) {
  static_assert(N % 16 == 0, "pad to a multiple of 16");
  (void)lane;
  {  // bit 5
    constexpr int H = N / 2;
#pragma unroll
    for (int n = 0; n < H; ++n) {
      double lo = v[n], hi = v[n + H];
      swap32_f64(lo, hi);
      v[n] = lo + hi;
    }
  }
  {  // bit 4
    constexpr int H = N / 4;
#pragma unroll
    for (int n = 0; n < H; ++n) {
      double lo = v[n], hi = v[n + H];
      swap16_f64(lo, hi);
      v[n] = lo + hi;
    }
  }
  {  // bit 3: partner = lane ^ 8 inside the 16-lane row (rotation by 8); banks 0,1 are bit-clear
    constexpr int H = N / 8;
#pragma unroll
    for (int n = 0; n < H; ++n) {
      double lo = v[n], hi = v[n + H];
      double recv = dpp_f64<kDppRowRor8, 0x3>(lo, lo);   // bit-clear lanes <- partner's lower value
      recv = dpp_f64<kDppRowRor8, 0xC>(recv, hi);        // bit-set lanes   <- partner's upper value
      double keep = dpp_f64<kDppQuadIdent, 0xC>(lo, hi);  // own lower / own upper
      v[n] = keep + recv;
    }
  }
  {  // bit 2: partner = lane ^ 4; banks 0,2 are bit-clear (take from lane+4 = ror 12), 1,3 bit-set
    constexpr int H = N / 16;
#pragma unroll
    for (int n = 0; n < H; ++n) {
      double lo = v[n], hi = v[n + H];
      double recv = dpp_f64<kDppRowRor12, 0x5>(lo, lo);
      recv = dpp_f64<kDppRowRor4, 0xA>(recv, hi);
      double keep = dpp_f64<kDppQuadIdent, 0xA>(lo, hi);
      v[n] = keep + recv;
    }
  }
  constexpr int R = N / 16;
#pragma unroll
  for (int n = 0; n < R; ++n) v[n] += dpp_f64<kDppQuadXor2, 0xF>(v[n], v[n]);
#pragma unroll
  for (int n = 0; n < R; ++n) v[n] += dpp_f64<kDppQuadXor1, 0xF>(v[n], v[n]);
}
// first original index owned by `lane` after wave_reduce_scatter<N>
template <int N>
__device__ __forceinline__ int reduce_scatter_base(int lane) {
  return (N / 16) * ((lane >> 2) & 15);
}

// Totals of up to four per-lane values over the 64 lanes in ONE butterfly (the class sums of an
// E phase): the two swap steps leave the partial sums of value r on row r of 16 lanes, four DPP
// steps inside the rows finish.  Returns the total of v_r on every lane of row r -- 7 cross-lane
// exchanges of doubles and 7 adds instead of the 6 + 6 per value of wave_sum.
__device__ __forceinline__ double wave_sum_rows4(double v0, double v1, double v2, double v3) {
  swap32_f64(v0, v2);
  double a = v0 + v2;  // lanes 0..31: v0, lanes 32..63: v2
  swap32_f64(v1, v3);
  double b = v1 + v3;  // lanes 0..31: v1, lanes 32..63: v3
  swap16_f64(a, b);
  double v = a + b;    // row 0: v0, row 1: v1, row 2: v2, row 3: v3
  v += dpp_f64<kDppQuadXor1, 0xF>(v, v);
  v += dpp_f64<kDppQuadXor2, 0xF>(v, v);
  v += dpp_f64<kDppRowHalfMirror, 0xF>(v, v);
  v += dpp_f64<kDppRowMirror, 0xF>(v, v);
  return v;
}
// red[k] = sum over the 64 lanes of s[k], k < K (K <= 8), written by one lane per class
template <int K>
__device__ __forceinline__ void wave_class_sums(const double (&s)[K], int lane, double* red) {
  static_assert(K <= 8, "two groups of four rows");
  const int row = lane >> 4;
  {
    const double t = wave_sum_rows4(s[0], K > 1 ? s[K > 1 ? 1 : 0] : 0.0, K > 2 ? s[K > 2 ? 2 : 0] : 0.0,
                                    K > 3 ? s[K > 3 ? 3 : 0] : 0.0);
    if ((lane & 15) == 0 && row < K) red[row] = t;
  }
  if constexpr (K > 4) {
    const double t = wave_sum_rows4(s[4], K > 5 ? s[K > 5 ? 5 : 0] : 0.0, K > 6 ? s[K > 6 ? 6 : 0] : 0.0,
                                    K > 7 ? s[K > 7 ? 7 : 0] : 0.0);
    if ((lane & 15) == 0 && row + 4 < K) red[row + 4] = t;
  }
}

// two all-reduce sums (independent instruction streams interleave)
__device__ __forceinline__ void wave_sum2(double& a, double& b) {
  a = wave_sum(a);
  b = wave_sum(b);
}

// acc += a16[lane N of this lane's row of 16] * y   (DP-ALU DPP, gfx90a+: a float64 FMA may take
// src0 through row_newbcast).  One VGPR pair whose 4 rows hold the same 16 values thereby serves
// as 16 wave-uniform operands -- cheaper than an LDS broadcast read or an SGPR per operand
// (tools/ubench/dpp_fmac.hip: same issue rate as the plain v_fmac_f64).  All 64 lanes must be
// active.  `a16` must not have been written by a VALU instruction in the two preceding issue
// slots (DPP read-after-VALU-write hazard; the operand registers are filled by LDS loads).
template <int N>
__device__ __forceinline__ void fmac_row_bcast(double& acc, double a16, double y) {
  static_assert(N >= 0 && N < 16, "lane within the row");
  asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
      : "+v"(acc)
      : "v"(a16), "v"(y), "n"(N));
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

// 1/sqrt(x) to ~1 ulp for finite normal x > 0: hardware estimate + two Newton steps
// (y <- y + y * (0.5 - 0.5 x y^2)), no division, no IEEE sqrt sequence.
__device__ __forceinline__ double fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double hx = 0.5 * x;
  double e = fma(-hx * y, y, 0.5);
  y = fma(y, e, y);
  e = fma(-hx * y, y, 0.5);
  y = fma(y, e, y);
  return y;
}

// exp(x) for x <= 0 (a max-shifted log-pdf): n = rint(x / ln 2), r = x - n ln 2 in two pieces,
// degree-12 Taylor polynomial on |r| <= 0.347 (truncation 1.7e-16), scaled by 2^n with v_ldexp
// (gradual underflow like libm's).  ~19 VALU instructions against the ~40 of the library call
// with its special-case handling; NaN propagates, and the lower clamp (written as a select so
// that a NaN survives it) keeps -inf away from the inf - inf of the range reduction.
__device__ __forceinline__ double exp_nonpos(double x) {
  x = (x < -800.0) ? -800.0 : x;
  const double n = __builtin_rint(x * 1.4426950408889634);
  double r = fma(-n, 0.6931471805599453094, x);
  r = fma(-n, 2.3190468138462996e-17, r);
  double p = 1.0 / 479001600.0;
  p = fma(p, r, 1.0 / 39916800.0);
  p = fma(p, r, 1.0 / 3628800.0);
  p = fma(p, r, 1.0 / 362880.0);
  p = fma(p, r, 1.0 / 40320.0);
  p = fma(p, r, 1.0 / 5040.0);
  p = fma(p, r, 1.0 / 720.0);
  p = fma(p, r, 1.0 / 120.0);
  p = fma(p, r, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)n);
}

// ln x for finite normal x > 0 and ln(1 + x) for |x| <= 0.35, ~2 ulp: mantissa / exponent split
// and the odd series of 2 atanh(z) (z = (m - 1)/(m + 1) resp. x/(2 + x), |z| <= 0.19: eleven terms
// leave 3e-18) on a Newton-refined reciprocal -- ~28 / ~20 instructions for the ~50 / ~70 of the
// library calls.  For the serial chains a whole workgroup waits for: ln c(kappa) of the Watson
// factorisation (-4.5 % of the 257-bin fit, profiles/r06_d_watson_phases.txt), the log-Bessel
// normaliser of the vMF model phase.
__device__ __forceinline__ double atanh2_series(double z) {
  const double z2 = z * z;
  double p = 1.0 / 21.0;
  p = fma(p, z2, 1.0 / 19.0);
  p = fma(p, z2, 1.0 / 17.0);
  p = fma(p, z2, 1.0 / 15.0);
  p = fma(p, z2, 1.0 / 13.0);
  p = fma(p, z2, 1.0 / 11.0);
  p = fma(p, z2, 1.0 / 9.0);
  p = fma(p, z2, 1.0 / 7.0);
  p = fma(p, z2, 1.0 / 5.0);
  p = fma(p, z2, 1.0 / 3.0);
  p = fma(p, z2, 1.0);
  return 2.0 * z * p;
}
__device__ __forceinline__ double log_pos(double x) {
  int e;
  double m = frexp(x, &e);  // [0.5, 1)
  const bool lo = m < 0.70710678118654752;
  m = lo ? 2.0 * m : m;     // [sqrt(1/2), sqrt(2)): the exponent term vanishes around x = 1
  e = lo ? e - 1 : e;
  const double z = (m - 1.0) * fast_rcp(m + 1.0);
  const double ed = (double)e;
  return fma(ed, 0.6931471805599453094, fma(ed, 2.3190468138462996e-17, atanh2_series(z)));
}
__device__ __forceinline__ double log1p_small(double x) {
  return atanh2_series(x * fast_rcp(2.0 + x));
}

constexpr double kTiny = 2.2250738585072014e-308;  // np.finfo(np.float64).tiny

}  // namespace pbbss
