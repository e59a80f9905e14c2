// Reference-precision instantiation of the persistent cACGMM EM kernel for gfx950 (MI355X):
// the E and M phases in PACKED FP32 (v_pk_fma_f32: two FMAs per lane and issue slot), class sums
// and the factorisation in float64.
//
// Why it exists.  For a complex64 observation with an ndarray initialisation the reference runs
// the WHOLE EM in complex64 / float32 (distribution/cacgmm.py:226-227: the affiliations are cast
// to the real dtype of Y, every einsum of cacg.py:185-199 / :316 then stays in single precision).
// The float64 kernel of cacgmm_em.hpp is a superset in accuracy but pays the FP64 vector rate --
// 16 FMA lanes per clock and SIMD, and a power-managed clock.  This kernel is the reference's own
// arithmetic for that case, labelled as such (pbbss_em_opts.precision = PBBSS_PRECISION_F32); the
// float64 kernel stays the default and the headline.
//
// Design (differs from the float64 kernel where the hardware does):
//  * Packing is over the (Re, Im) pair of one Hermitian entry, not over two frames: with
//    y_i = (re_i, im_i) in one 64-bit VGPR pair -- the natural complex64 layout --
//        P_ij = y_i conj(y_j) = (re_i re_j + im_i im_j, im_i re_j - re_i im_j)
//    is ONE v_pk_mul_f32 + ONE v_pk_fma_f32 (op_sel picks / swaps the halves, neg_hi flips one
//    sign), the quadratic form takes (2 Re A_ij, 2 Im A_ij) as ONE 64-bit operand:
//        (qa, qb) += (2 Re A, 2 Im A) * (Re P, Im P),      q = qa + qb,
//    and the covariance update keeps (Re C_ij, Im C_ij) in one accumulator pair:
//        (Re C, Im C) += w * (Re P, Im P).
//    Lane = frame in both phases, half the accumulator registers of a frame-packed layout.
//  * Operand feed of the E phase: A_k is wave-uniform.  VOP3P has no DPP (the float64 kernel's
//    row_newbcast trick does not exist for packed math).  After the factorisation the wave that
//    owns class k leaves A_k as float32 in LDS (K x 64 floats), and the E phase fetches it with
//    broadcast ds_read_b128 (all lanes one address: conflict-free, four operands per read) in
//    chunks of eight operands, the next chunk in flight under the FMAs of the current one; every
//    operand pair feeds two frames per lane.  48 reads per wave and iteration keep the LDS pipe
//    at about a quarter of its capacity.  (First version: SGPR operands streamed with
//    s_load_dwordx8 ... glc from an L2-resident slot -- correct, and the natural home of uniform
//    operands, but scalar loads return out of order, so only lgkmcnt(0) can be waited for and
//    the L2 round trip of every chunk was exposed: the E phase took 3x the M phase,
//    profiles/r03_b_f32_experiments.txt.)
//  * y is kept PRE-NORMALISED in LDS (float32, 32 KB at T = 500, D = 8): no widening, no
//    1 / |y|^2 factor anywhere; M-step weights as float32 (K padded to 2, 4 or 8 per frame, one
//    ds_write_b128 / ds_read_b128 per frame).
//  * Softmax in mantissa / exponent form as in the float64 kernel, on v_frexp / v_rcp_f32.
//  * The M-phase butterfly runs on 32-bit registers (half the swaps), the totals are widened
//    once and from there on the float64 code of cacgmm_em.hpp is used as is: class sums,
//    Gauss-Jordan / Jacobi factorisation with the reference's floor, mixture weights, status
//    bits, and the split-group protocol for the remainder bins (member workgroups in the grid).
#pragma once
#include "cacgmm_em.hpp"

namespace pbbss {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

constexpr float kTiny32 = 1.17549435e-38f;  // np.finfo(np.float32).tiny

// ---- packed-FP32 building blocks ------------------------------------------------------------
// acc += w[HI] * p: both halves take the same dword of the VGPR pair w (per-lane M-step weights of
// two classes in one pair; a broadcast-read operand pair of the E phase)
template <int HI>
__device__ __forceinline__ void pkfma_v_bcast(f32x2& acc, f32x2 w, f32x2 p) {
  if constexpr (HI == 0) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(w), "v"(p));
  } else {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(w), "v"(p));
  }
}
// (Re, Im) of y_i conj(y_j) from yi = (re_i, im_i), yj = (re_j, im_j): two packed instructions
__device__ __forceinline__ f32x2 herm_pair(f32x2 yi, f32x2 yj) {
  f32x2 t, p;
  // t = (re_i re_j, im_i re_j)
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(yi), "v"(yj));
  // p = (im_i im_j + t.lo, -re_i im_j + t.hi)
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]"
      : "=v"(p)
      : "v"(yi), "v"(yj), "v"(t));
  return p;
}
template <int H>
__device__ __forceinline__ f32x2 pair_of(f32x8 v) {
  return __builtin_shufflevector(v, v, 2 * H, 2 * H + 1);
}
template <int H>
__device__ __forceinline__ f32x2 pair_of(f32x4 v) {
  return __builtin_shufflevector(v, v, 2 * H, 2 * H + 1);
}

// ---- 32-bit cross-lane helpers (wave_reduce_scatter of pbbss_dev.hpp on single registers) -----
template <int CTRL, int BANK>
__device__ __forceinline__ float dpp_f32(float old, float src) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src),
                                         CTRL, 0xF, BANK, false));
}
__device__ __forceinline__ void swap32_f32(float& a, float& b) {
  auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a),
                                            __builtin_bit_cast(unsigned, b), false, false);
  a = __builtin_bit_cast(float, (unsigned)r[0]);
  b = __builtin_bit_cast(float, (unsigned)r[1]);
}
__device__ __forceinline__ void swap16_f32(float& a, float& b) {
  auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a),
                                            __builtin_bit_cast(unsigned, b), false, false);
  a = __builtin_bit_cast(float, (unsigned)r[0]);
  b = __builtin_bit_cast(float, (unsigned)r[1]);
}
// Same contract as wave_reduce_scatter<N>(double (&)[N], lane): v[0 .. N/16) = totals of the
// original indices base + m, base = (N/16) * ((lane >> 2) & 15).
template <int N>
__device__ __forceinline__ void wave_reduce_scatter32(float (&v)[N]) {
  static_assert(N % 16 == 0, "pad to a multiple of 16");
  {
    constexpr int H = N / 2;
#pragma unroll
    for (int n = 0; n < H; ++n) {
      float lo = v[n], hi = v[n + H];
      swap32_f32(lo, hi);
      v[n] = lo + hi;
    }
  }
  {
    constexpr int H = N / 4;
#pragma unroll
    for (int n = 0; n < H; ++n) {
      float lo = v[n], hi = v[n + H];
      swap16_f32(lo, hi);
      v[n] = lo + hi;
    }
  }
  {
    constexpr int H = N / 8;
#pragma unroll
    for (int n = 0; n < H; ++n) {
      float lo = v[n], hi = v[n + H];
      float recv = dpp_f32<kDppRowRor8, 0x3>(lo, lo);
      recv = dpp_f32<kDppRowRor8, 0xC>(recv, hi);
      float keep = dpp_f32<kDppQuadIdent, 0xC>(lo, hi);
      v[n] = keep + recv;
    }
  }
  {
    constexpr int H = N / 16;
#pragma unroll
    for (int n = 0; n < H; ++n) {
      float lo = v[n], hi = v[n + H];
      float recv = dpp_f32<kDppRowRor12, 0x5>(lo, lo);
      recv = dpp_f32<kDppRowRor4, 0xA>(recv, hi);
      float keep = dpp_f32<kDppQuadIdent, 0xA>(lo, hi);
      v[n] = keep + recv;
    }
  }
  constexpr int R = N / 16;
#pragma unroll
  for (int n = 0; n < R; ++n) v[n] += dpp_f32<kDppQuadXor2, 0xF>(v[n], v[n]);
#pragma unroll
  for (int n = 0; n < R; ++n) v[n] += dpp_f32<kDppQuadXor1, 0xF>(v[n], v[n]);
}

template <int D, int K>
struct EmKernel32 {
  static_assert(K >= 1 && K <= 6, "packed-FP32 instantiation: up to six classes");
  using Base = EmKernel<D, K, float, false>;
  using BLds = typename Base::Lds;
  static constexpr int DP = Base::DP, NOFF = Base::NOFF, NA = Base::NA;
  static constexpr int NDW = Base::NDW, NOW = Base::NOW, NSLOT = Base::NSLOT, NACC = Base::NACC;
  static constexpr int DPAD = (D + 1) & ~1;                 // diagonal block of the operand row, even
  static constexpr int NA32 = DPAD + 2 * NOFF;              // operands of one class
  static constexpr int NAP = (NA32 + 7) & ~7;               // row stride: whole chunks of eight operands
  static constexpr int NCH = NAP / 8;
  static constexpr int KP = K <= 2 ? 2 : (K <= 4 ? 4 : 8);  // M-step weights of a frame: one or two vectors

  struct Lds {
    BLds b;     // the float64 small arrays of the float64 kernel (frame arrays unused)
    float* y;   // [DP][Tp][4]  pre-normalised channel pairs (re_a, im_a, re_b, im_b), frame contiguous
    float* w;   // [Tp][KP]     M-step weights gamma / q
    float* a32; // [K][NAP]     A_k as float32 operand rows: [diagonal (DPAD) | (2 Re, 2 Im) pairs]
    int Tp;
  };

  static __host__ __device__ size_t frame_bytes(int T) {
    const size_t Tp = (size_t)((T + 1) & ~1);
    return (size_t)DP * Tp * 16 + Tp * KP * 4;
  }
  static __host__ __device__ size_t lds_bytes(int T) {
    return (Base::small_bytes() + frame_bytes(T) + (size_t)K * NAP * 4 + 15) & ~(size_t)15;
  }

  static __device__ Lds carve(char* base, int T) {
    Lds L;
    L.Tp = (T + 1) & ~1;
    L.y = reinterpret_cast<float*>(base);
    L.w = L.y + (size_t)DP * L.Tp * 4;
    L.a32 = L.w + (size_t)L.Tp * KP;  // Tp even, KP even: 16-byte aligned when Tp * KP % 4 == 0
    L.b = Base::carve_small(reinterpret_cast<char*>(L.a32 + (size_t)K * NAP), L.Tp);
    Base::template fill_wbtab<true>(L.b);  // write-back table of phase_m (this kernel's entry map)
    return L;
  }

  template <int I>
  static __device__ __forceinline__ f32x2 chan(const f32x4 (&yv)[DP]) {
    return pair_of<I % 2>(yv[I / 2]);
  }
  static __device__ __forceinline__ void load_frame(const Lds& L, int t, f32x4 (&yv)[DP]) {
    PBBSS_DEV_ASSERT(t >= 0 && t < L.Tp);
#pragma unroll
    for (int dp = 0; dp < DP; ++dp)
      yv[dp] = *reinterpret_cast<const f32x4*>(L.y + ((size_t)dp * L.Tp + t) * 4);
  }

  // ---- phase L: HBM -> LDS, unit-normalised in float32 (utils.py:223-256, input precision) ----
  static __device__ void phase_load(const EmArgs& a, const Lds& L, int64_t b, int tid, int tf = 0) {
    const int T = a.T;
    const int TS = Base::t_stride(a);
    const float2* yg = reinterpret_cast<const float2*>(a.y);
    bool zero_seen = false;
    for (int t = tid; t < L.Tp; t += kEmThreads) {
      float vr[2 * DP], vi[2 * DP];
#pragma unroll
      for (int d = 0; d < 2 * DP; ++d) {
        vr[d] = 0.f;
        vi[d] = 0.f;
      }
      float n2 = 0.f;
      if (t < T) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
          const size_t idx = (a.layout == PBBSS_LAYOUT_TD) ? ((size_t)b * TS + tf + t) * D + d
                                                           : ((size_t)b * D + d) * TS + tf + t;
          const float2 v = yg[idx];
          vr[d] = v.x;
          vi[d] = v.y;
          n2 = fmaf(v.x, v.x, fmaf(v.y, v.y, n2));
        }
      }
      float inv = 1.f;  // PBBSS_LAYOUT_DT: the caller already normalised
      if (a.layout == PBBSS_LAYOUT_TD) inv = (n2 > 0.f) ? 1.f / sqrtf(n2) : 0.f;
#pragma unroll
      for (int dp = 0; dp < DP; ++dp) {
        f32x4 o = {vr[2 * dp] * inv, vi[2 * dp] * inv, vr[2 * dp + 1] * inv, vi[2 * dp + 1] * inv};
        *reinterpret_cast<f32x4*>(L.y + ((size_t)dp * L.Tp + t) * 4) = o;
      }
      if (t < T && !(n2 > 0.f)) zero_seen = true;
    }
    // an all-zero frame pins the problem to the exact eigen path (see cacgmm_em.hpp: phase_load)
    if (zero_seen) atomicOr(L.b.flags, 1);
  }

  static __device__ __forceinline__ void store_weights(const Lds& L, int t, const float (&w)[K]) {
    if constexpr (KP == 2) {
      f32x2 o = {w[0], K > 1 ? w[K > 1 ? 1 : 0] : 0.f};
      *reinterpret_cast<f32x2*>(L.w + (size_t)t * KP) = o;
    } else if constexpr (KP == 4) {
      f32x4 o = {w[0], w[1], w[2], K > 3 ? w[K > 3 ? 3 : 0] : 0.f};
      *reinterpret_cast<f32x4*>(L.w + (size_t)t * KP) = o;
    } else {
      f32x4 o0 = {w[0], w[1], w[2], w[3]};
      f32x4 o1 = {w[K > 4 ? 4 : 0], K > 5 ? w[K > 5 ? 5 : 0] : 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(L.w + (size_t)t * KP) = o0;
      *reinterpret_cast<f32x4*>(L.w + (size_t)t * KP + 4) = o1;
    }
  }

  // ---- phase I: weights from an affiliation initialisation (cacgmm.py:211-228, q = 1) ----
  static __device__ void phase_init_gamma(const EmArgs& a, const Lds& L, int64_t b, int tid,
                                          int wave, int lane, int tf = 0) {
    const int TS = Base::t_stride(a);
    double s[K];
#pragma unroll
    for (int k = 0; k < K; ++k) s[k] = 0.0;
    for (int t = tid; t < a.T; t += kEmThreads) {
      const double sal = a.saliency ? a.saliency[(size_t)b * TS + tf + t] : 1.0;
      float w[K];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const size_t idx = ((size_t)b * K + k) * TS + tf + t;
        const double g = a.gamma0[idx] * sal;
        const double q = a.q0 ? a.q0[idx] : 1.0;
        w[k] = (float)(g / fmax(q, 10.0 * (double)kTiny32));
        s[k] += g;
      }
      store_weights(L, t, w);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const double tot = wave_sum(s[k]);
      if (lane == 0) L.b.red[wave * K + k] = tot;
    }
  }

  // ---- phase E ---------------------------------------------------------------------------------
  // MASK: source_activity_mask given (its own instantiation: a wave-uniform test inside the
  // softmax block cost the mask-less fit 2 % -- it splits the block the scheduler overlaps the
  // next operand chunk with)
  template <bool FINAL, bool MASK>
  static __device__ void phase_e_impl(const EmArgs& a, const Lds& L, int64_t b, int tid, int wave,
                                 int lane, float eps, int tf = 0) {
    tid = opaque(tid);
    lane = opaque(lane);
    const int TS = Base::t_stride(a);
    float s[K];
#pragma unroll
    for (int k = 0; k < K; ++k) s[k] = 0.f;
    auto pass = [&](int t0, auto nfc) {
      constexpr int NF = decltype(nfc)::value;  // frames per lane: every operand feeds NF frames
      int tt[NF];
      bool ok[NF];
      f32x4 yv[NF][DP];
      f32x2 q2[NF][K];
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const int t = t0 + f * kEmThreads + tid;
        ok[f] = t < a.T;
        tt[f] = ok[f] ? t : (a.T - 1);
        load_frame(L, tt[f], yv[f]);
#pragma unroll
        for (int k = 0; k < K; ++k) q2[f][k] = f32x2{0.f, 0.f};
      }
      // operand stream: broadcast reads of chunk c + 1 are issued BEFORE the FMAs of chunk c (LDS
      // returns in order, the compiler's lgkmcnt bookkeeping overlaps them); scheduling barriers
      // pin that order, the partial sums are tied to the stage boundary (cacgmm_em.hpp, code
      // generation note iv: pure arithmetic is not ordered by memory clobbers)
      const f32x4* a4 = reinterpret_cast<const f32x4*>(L.a32);
      f32x4 ch[2][K][2];
      auto fetch = [&](auto cc, auto bb) {
        constexpr int c = cc, bsel = bb;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          ch[bsel][k][0] = a4[(k * NAP + 8 * c) / 4];
          ch[bsel][k][1] = a4[(k * NAP + 8 * c) / 4 + 1];
        }
      };
      fetch(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
      asm volatile("" ::: "memory");
      static_for<0, NCH>([&](auto cc) {
        constexpr int c = cc;
        constexpr int cur = c & 1;
        if constexpr (c + 1 < NCH)
          fetch(std::integral_constant<int, c + 1>{}, std::integral_constant<int, 1 - cur>{});
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, 4>([&](auto hc) {
          constexpr int h = hc;            // operand pair h of the chunk
          constexpr int e = 8 * c + 2 * h;  // first operand index of the pair
          if constexpr (e < DPAD) {
            // two diagonal operands A_ii, A_(i+1)(i+1): each scales (re_i^2, im_i^2)
            static_for<0, 2>([&](auto uc) {
              constexpr int u = uc;
              constexpr int i = e + u;
              if constexpr (i < D) {
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                  const f32x2 yi = chan<i>(yv[f]);
                  const f32x2 dg = yi * yi;
#pragma unroll
                  for (int k = 0; k < K; ++k)
                    pkfma_v_bcast<u>(q2[f][k], pair_of<h % 2>(ch[cur][k][h / 2]), dg);
                }
              }
            });
          } else if constexpr (e < NA32) {
            constexpr int p = (e - DPAD) / 2;
            constexpr int i = tri_i<D>(p), j = tri_j<D>(p);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
              const f32x2 P = herm_pair(chan<i>(yv[f]), chan<j>(yv[f]));
#pragma unroll
              for (int k = 0; k < K; ++k)
                q2[f][k] = __builtin_elementwise_fma(pair_of<h % 2>(ch[cur][k][h / 2]), P, q2[f][k]);
            }
          }
        });
#pragma unroll
        for (int f = 0; f < NF; ++f) {
#pragma unroll
          for (int k = 0; k < K; ++k) asm volatile("" : "+v"(q2[f][k])::"memory");
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      // source_activity_mask of the EM loop / of the final predict (wave-uniform pointer)
      const uint8_t* act = FINAL ? a.final_activity : a.activity;
      // per-class constants of the softmax (float64 in LDS, written by the factorisation)
      float rdet[K], wgt[K];
      int dete[K];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        rdet[k] = (float)L.b.rdet[k];
        dete[k] = L.b.dete[k];
        wgt[k] = (float)L.b.wgt[k];
      }
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const int t = tt[f];
        // exp(log_pdf_k) = 1 / (det_k q_k^D) in mantissa / exponent form (cacg.py:200-201)
        float q[K], val[K], rq[K];
        int ex[K];
        int emax = INT32_MIN;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const float qq = fmaxf(fabsf(q2[f][k].x + q2[f][k].y), kTiny32);  // cacg.py:185-199
          q[k] = qq;
          const int e = __builtin_amdgcn_frexp_expf(qq);
          const float m = __builtin_amdgcn_frexp_mantf(qq);  // [0.5, 1)
          const float rm = __builtin_amdgcn_rcpf(m);
          rq[k] = ldexpf(rm, -e);
          float pw = rm * rm;  // rm^D
          if constexpr (D == 2) {
          } else if constexpr (D == 3) {
            pw = pw * rm;
          } else if constexpr (D == 4) {
            pw = pw * pw;
          } else if constexpr (D == 5) {
            pw = pw * pw * rm;
          } else if constexpr (D == 6) {
            pw = pw * pw * pw;
          } else if constexpr (D == 7) {
            pw = pw * pw * pw * rm;
          } else {
            pw = pw * pw;
            pw = pw * pw;
          }
          val[k] = rdet[k] * pw;
          ex[k] = -(e * D + dete[k]);
          emax = max(emax, ex[k]);
        }
        // weight x source_activity_mask (:39-41)
        float wam[K];
#pragma unroll
        for (int k = 0; k < K; ++k) wam[k] = wgt[k];
        if constexpr (MASK) {
#pragma unroll
          for (int k = 0; k < K; ++k) wam[k] *= (float)act[((size_t)b * K + k) * TS + tf + t];
        }
        float g[K], den = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const float v = ldexpf(val[k], ex[k] - emax) * wam[k];  // mixture_model_utils.py:32-41
          g[k] = v;
          den += v;
        }
        // 0, or NaN for a non-finite class sum: v_max / v_min drop the NaN that np.maximum / np.clip
        // keep -- it rides on the saliency factor into the weights and the class sums, so that the
        // covariance turns non-finite and the status says so (cacgmm_em.hpp, same place)
        const float poison = den - den;
        den = fmaxf(den, kTiny32);  // :43-47
        const float rden = __builtin_amdgcn_rcpf(den);
        const float sal =
            ((!FINAL && a.saliency) ? (float)a.saliency[(size_t)b * TS + tf + t] : 1.f) + poison;
        float wout[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
          float gam = g[k] * rden;
          // :50-53, no renormalisation; unconditional inside the loop (as in cacgmm_em.hpp: with
          // eps = 0 the clip is [0, 1])
          if (!FINAL || eps != 0.f) gam = fminf(fmaxf(gam, eps), 1.f - eps);
          if constexpr (FINAL) {
            if (ok[f] && a.out_aff)
              a.out_aff[((size_t)b * K + k) * TS + tf + t] = (double)(gam + poison);
            wout[k] = 0.f;
          } else {
            const float gs = ok[f] ? gam * sal : 0.f;
            // gamma / max(q, 10 tiny) (cacg.py:310, :322); y is unit-norm in LDS
            const float rqk = fminf(rq[k], 1.f / (10.f * kTiny32));  // one v_min instead of compare + select
            wout[k] = gs * rqk;
            s[k] += gs;
          }
        }
        if constexpr (!FINAL) {
          if (ok[f]) store_weights(L, t, wout);
        }
      }
    };
    {
      int t0 = 0;
      for (; a.T - t0 > kEmThreads; t0 += 2 * kEmThreads) pass(t0, std::integral_constant<int, 2>{});
      for (; t0 < a.T; t0 += kEmThreads) pass(t0, std::integral_constant<int, 1>{});
    }
    if constexpr (!FINAL) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const double tot = wave_sum((double)s[k]);
        if (lane == 0) L.b.red[wave * K + k] = tot;
      }
    }
  }

  template <bool FINAL>
  static __device__ __forceinline__ void phase_e(const EmArgs& a, const Lds& L, int64_t b, int tid,
                                                 int wave, int lane, float eps, int tf = 0) {
    if ((FINAL ? a.final_activity : a.activity) != nullptr) {
      phase_e_impl<FINAL, true>(a, L, b, tid, wave, lane, eps, tf);
    } else {
      phase_e_impl<FINAL, false>(a, L, b, tid, wave, lane, eps, tf);
    }
  }

  // ---- phase M: wave W accumulates its share of the Hermitian entries (as cacgmm_em.hpp) -------
  static constexpr int NPK = NDW + NOW;  // packed accumulators per class and wave
  template <int W>
  static __device__ void phase_m(const EmArgs& a, const Lds& L, int lane) {
    lane = opaque(lane);
    f32x2 acc[K * NPK];
#pragma unroll
    for (int x = 0; x < K * NPK; ++x) acc[x] = f32x2{0.f, 0.f};
    // One trip = 64 frames (lane = frame); frames beyond T are read at a clamped address with their
    // weights zeroed (no guarded loads).  (A software-pipelined version -- ping-pong operand
    // sets, the loads of trip n + 1 issued before the FMAs of trip n -- measured 3 % SLOWER on
    // one box, profiles/r03_b_f32_experiments.txt: two workgroups per CU already cover the LDS
    // latency, the extra live registers and scheduling barriers cost more.)
    struct Trip {
      f32x4 yv[DP];
      f32x2 wp[KP / 2];
    };
    auto fetch = [&](Trip& tr, int t0) {
      const int t = t0 + lane;
      const bool ok = t < a.T;
      const int tc = ok ? t : a.T - 1;
      load_frame(L, tc, tr.yv);
      if constexpr (KP == 2) {
        tr.wp[0] = *reinterpret_cast<const f32x2*>(L.w + (size_t)tc * KP);
      } else {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(L.w + (size_t)tc * KP);
        tr.wp[0] = pair_of<0>(wv);
        tr.wp[1] = pair_of<1>(wv);
        if constexpr (KP == 8) {
          const f32x4 wv1 = *reinterpret_cast<const f32x4*>(L.w + (size_t)tc * KP + 4);
          tr.wp[2] = pair_of<0>(wv1);
          tr.wp[3] = pair_of<1>(wv1);
        }
      }
#pragma unroll
      for (int x = 0; x < KP / 2; ++x) tr.wp[x] = ok ? tr.wp[x] : f32x2{0.f, 0.f};
    };
    auto accumulate = [&](const Trip& tr) {
      // all Hermitian products of the trip first (independent two-instruction chains the scheduler
      // can interleave), then the K x (diagonals + pairs) multiply-adds
      f32x2 dgv[NDW > 0 ? NDW : 1], pv[NOW > 0 ? NOW : 1];
      static_for<0, D>([&](auto ic) {
        constexpr int i = ic;
        if constexpr (i % kEmWaves == W) {
          const f32x2 yi = chan<i>(tr.yv);
          dgv[i / kEmWaves] = yi * yi;
        }
      });
      static_for<0, NOFF>([&](auto pc) {
        constexpr int p = pc;
        if constexpr (p % kEmWaves == W) {
          constexpr int i = tri_i<D>(p), j = tri_j<D>(p);
          pv[p / kEmWaves] = herm_pair(chan<i>(tr.yv), chan<j>(tr.yv));
        }
      });
      static_for<0, D>([&](auto ic) {
        constexpr int i = ic;
        if constexpr (i % kEmWaves == W) {
          static_for<0, K>([&](auto kc) {
            constexpr int k = kc;
            pkfma_v_bcast<k % 2>(acc[k * NPK + i / kEmWaves], tr.wp[k / 2], dgv[i / kEmWaves]);
          });
        }
      });
      static_for<0, NOFF>([&](auto pc) {
        constexpr int p = pc;
        if constexpr (p % kEmWaves == W) {
          static_for<0, K>([&](auto kc) {
            constexpr int k = kc;
            pkfma_v_bcast<k % 2>(acc[k * NPK + NDW + p / kEmWaves], tr.wp[k / 2], pv[p / kEmWaves]);
          });
        }
      });
    };
    for (int t0 = 0; t0 < a.T; t0 += kWave) {
      Trip tr;
      fetch(tr, t0);
      accumulate(tr);
    }
    // flatten to the slot order of the float64 kernel: per class NDW diagonals, then (Re, Im) pairs
    float flat[NACC];
#pragma unroll
    for (int x = 0; x < NACC; ++x) flat[x] = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
#pragma unroll
      for (int s = 0; s < NDW; ++s) flat[k * NSLOT + s] = acc[k * NPK + s].x + acc[k * NPK + s].y;
#pragma unroll
      for (int s = 0; s < NOW; ++s) {
        flat[k * NSLOT + NDW + 2 * s] = acc[k * NPK + NDW + s].x;
        flat[k * NSLOT + NDW + 2 * s + 1] = acc[k * NPK + NDW + s].y;
      }
    }
    // issue priority of the dependent chains beside the co-resident workgroup's stream
    // (cacgmm_em.hpp: phase_m); member workgroups keep the level of their launch
    const bool host = a.main_grid <= 0 || (int)blockIdx.x < a.main_grid;
    if (host) __builtin_amdgcn_s_setprio(1);
    wave_reduce_scatter32<NACC>(flat);
    if ((lane & 3) == 0) {  // destinations from the table carve() left in LDS (cacgmm_em.hpp: fill_wbtab)
      const unsigned short* tab = L.b.wbtab + (W * 16 + ((lane >> 2) & 15)) * Base::kWbR;
#pragma unroll
      for (int m = 0; m < Base::kWbR; ++m) L.b.cpack[tab[m]] = (double)flat[m];
    }
    if (host) __builtin_amdgcn_s_setprio(0);
  }
  static __device__ __forceinline__ void phase_m_dispatch(const EmArgs& a, const Lds& L, int wave,
                                                          int lane) {
    switch (wave) {
      case 0: phase_m<0>(a, L, lane); break;
      case 1: phase_m<1>(a, L, lane); break;
      case 2: phase_m<2>(a, L, lane); break;
      default: phase_m<3>(a, L, lane); break;
    }
  }

  // ---- A_k (float64, LDS, packed as apack) -> float32 operand row of class k (LDS) -------------
  static __device__ __forceinline__ void publish_class(const Lds& L, int k, int lane) {
    for (int i = lane; i < NA; i += kWave) {
      const int i32 = (i < D) ? i : i + (DPAD - D);
      L.a32[k * NAP + i32] = (float)L.b.apack[k * NA + i];
    }
  }
  static __device__ __forceinline__ void publish_all(const Lds& L, int tid) {
    for (int x = tid; x < K * NA; x += kEmThreads) {
      const int k = x / NA, i = x - k * NA;
      const int i32 = (i < D) ? i : i + (DPAD - D);
      L.a32[k * NAP + i32] = (float)L.b.apack[x];
    }
  }

  // ---- member workgroup of a remainder problem (run_split of cacgmm_em.hpp, packed phases) -----
  static __device__ void run_member(const EmArgs& ga, char* smem, int mblock, int nblocks) {
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int G = ga.split_groups;
    const int prob = mblock / G, g = mblock % G;
    const int nprob = nblocks / G;
    const int64_t b = ga.b_first + prob;
    const int tf = g * ga.split_window;
    switch (ga.split_prio) {
      case 1: __builtin_amdgcn_s_setprio(1); break;
      case 2: __builtin_amdgcn_s_setprio(2); break;
      case 3: __builtin_amdgcn_s_setprio(3); break;
      default: break;
    }
    EmArgs a = ga;  // this workgroup's window
    a.T = min(ga.split_window, ga.T_total - tf);
    const Lds L = carve(smem, ga.split_window);
    if (tid < K) {
      L.b.status[tid] = 0;
      if (g == 0 && a.out_status) a.out_status[(size_t)b * K + tid] = 0;
    }
    if (tid == 0) *L.b.flags = 0;
    __syncthreads();
    phase_load(a, L, b, tid, tf);
    __syncthreads();
    const bool model_in = (a.gamma0 == nullptr);
    if (model_in) {
      for (int k = wave; k < K; k += kEmWaves) Base::prep_from_model(a, L.b, b, k, lane);
      __syncthreads();
      publish_all(L, tid);
    } else {
      phase_init_gamma(a, L, b, tid, wave, lane, tf);
    }
    __syncthreads();
    for (int it = 0; it < a.iterations; ++it) {
      if (it > 0 || model_in) {
        if ((wave << 6) < a.T) {  // windows are <= 256 frames: one E pass
          phase_e<false>(a, L, b, tid, wave, lane, (float)a.aff_eps, tf);
        } else if (lane < K) {
          L.b.red[wave * K + lane] = 0.0;
        }
        __syncthreads();
      }
      phase_m_dispatch(a, L, wave, lane);
      __syncthreads();
      Base::split_exchange(a, L.b, prob, nprob, g, it, tid);
      const bool last = (it == a.iterations - 1);
      if (G >= K) {
        if (g < K && wave == kEmWaves - 1) Base::factor_class(a, L.b, b, g, lane, last);
        __syncthreads();
        Base::split_publish_model(a, L.b, prob, nprob, g, it, tid);
      } else {
        EmArgs fa = a;
        if (g != 0) {
          fa.out_eigvec = nullptr;
          fa.out_eigval = nullptr;
          fa.out_cov = nullptr;
        }
        for (int k = wave; k < K; k += kEmWaves) Base::factor_class(fa, L.b, b, k, lane, last);
        __syncthreads();
      }
      publish_all(L, tid);
      __syncthreads();
    }
    if (tid < K) {
      if (g == 0 && a.out_weight && a.iterations > 0) a.out_weight[(size_t)b * K + tid] = L.b.wgt[tid];
      const int bits = (g == 0 ? L.b.status[tid] : 0) |
                       (Base::split_failed(a) ? (PBBSS_ST_EIG_NOCONV | PBBSS_ST_NONFINITE) : 0);
      if (a.out_status && bits) atomicOr(a.out_status + (size_t)b * K + tid, bits);
    }
    if (a.final_predict) phase_e<true>(a, L, b, tid, wave, lane, (float)a.final_eps, tf);
    if (tid == 0) {  // the member that leaves last zeroes the arrival counters of the problem
      unsigned* ex = a.xcount + 8 + prob;
      const unsigned before =
          __hip_atomic_fetch_add(ex, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (before == (unsigned)(G - 1)) {
        __hip_atomic_store(a.xcount + prob, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.xcount + 16 + prob, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(ex, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }

  static __device__ void run(const EmArgs& a, char* smem) {
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    if (a.main_grid > 0 && (int)blockIdx.x >= a.main_grid) {
      run_member(a, smem, (int)blockIdx.x - a.main_grid, (int)gridDim.x - a.main_grid);
      return;
    }
    const int bstride = a.main_grid > 0 ? a.main_grid : (int)gridDim.x;
    PBBSS_DEV_ASSERT(a.lds_given == 0 || lds_bytes(a.T) <= a.lds_given);
    const Lds L = carve(smem, a.T);
#ifdef PBBSS_PHASE_PROFILE
    unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_readcyclecounter();
#define PBBSS_TICK32(i)                                      \
  {                                                          \
    unsigned long long tn = __builtin_readcyclecounter();    \
    pc[i] += tn - tprev;                                     \
    tprev = tn;                                              \
  }
#else
#define PBBSS_TICK32(i)
#endif
    for (int64_t b = blockIdx.x; b < a.B; b += bstride) {
      __syncthreads();  // previous problem fully retired before LDS is reused
      if (tid < K) L.b.status[tid] = 0;
      if (tid == 0) *L.b.flags = 0;
      __syncthreads();
      phase_load(a, L, b, tid);
      __syncthreads();
      const bool model_in = (a.gamma0 == nullptr);
      if (model_in) {
        for (int k = wave; k < K; k += kEmWaves) {
          Base::prep_from_model(a, L.b, b, k, lane);
          publish_class(L, k, lane);
        }
      } else {
        phase_init_gamma(a, L, b, tid, wave, lane);
      }
      __syncthreads();
      PBBSS_TICK32(0)
      for (int it = 0; it < a.iterations; ++it) {
        if (it > 0 || model_in) {
          phase_e<false>(a, L, b, tid, wave, lane, (float)a.aff_eps);
          PBBSS_TICK32(1)
          __syncthreads();
          PBBSS_TICK32(2)
        }
        phase_m_dispatch(a, L, wave, lane);
        PBBSS_TICK32(3)
        __syncthreads();
        PBBSS_TICK32(4)
        const bool last = (it == a.iterations - 1);
        // (factor_class raises the priority itself only where split_groups == 0; this launch
        // carries the group count for its in-grid members, so the hosts do it here)
        __builtin_amdgcn_s_setprio(3);
        for (int k = wave; k < K; k += kEmWaves) {
          Base::factor_class(a, L.b, b, k, lane, last);
          publish_class(L, k, lane);  // same wave wrote apack of class k: wave-ordered LDS
        }
        __builtin_amdgcn_s_setprio(0);
        PBBSS_TICK32(5)
        __syncthreads();
        PBBSS_TICK32(6)
      }
      if (tid < K) {
        if (a.out_weight && a.iterations > 0) a.out_weight[(size_t)b * K + tid] = L.b.wgt[tid];
        if (a.out_status) a.out_status[(size_t)b * K + tid] = L.b.status[tid];
      }
      if (a.final_predict) phase_e<true>(a, L, b, tid, wave, lane, (float)a.final_eps);
      PBBSS_TICK32(7)
    }
#ifdef PBBSS_PHASE_PROFILE
    // [wave][8] cycle sums over all workgroups (same slots as the float64 kernel)
    if (a.prof && lane == 0) {
      for (int i = 0; i < 8; ++i) atomicAdd(a.prof + wave * 8 + i, pc[i]);
    }
#endif
  }
};

template <int D, int K>
__global__ void __launch_bounds__(kEmThreads, em_waves_per_simd(K)) cacgmm_em32_kernel(EmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  EmKernel32<D, K>::run(a, smem);
}

}  // namespace pbbss
