// Persistent complex-Watson mixture-model (cWMM) EM kernel for gfx950
// (BASELINE config 4).
//
// Reference: distribution/cwmm.py:151-240 (CWMMTrainer._fit / _m_step),
// :25-52 (CWMM.predict), distribution/complex_watson.py:73-87 (log_pdf),
// :157-168 (log_norm_1f1), :238-271 (concentration from the largest eigenvalue
// through a quadratic spline of the inverse hypergeometric ratio), :300-315
// (_fit: weighted covariance -> dominant eigenvector).
//
// Same skeleton as the cACGMM kernel (cacgmm_em.hpp), whose staging, outer
// product / quadratic form and entry-split M phase are reused verbatim:
//   E  q_kt = |m_k^H y_t|^2 = <m_k m_k^H, P_t>;  log p = kappa_k q_kt - ln c(kappa_k);
//      softmax with a max-shift (float64 exp), weights in the linear domain
//   M  C_k = sum_t gamma_kt P_t / sum_t gamma_kt      (no 1/q weighting)
//   F  wave k: Jacobi eigendecomposition -> (lambda_max, m_k); kappa_k by de Boor
//      evaluation of the host-built quadratic B-spline (the host builds it with
//      SciPy exactly as the reference does and passes knots + coefficients);
//      ln c(kappa) = ln(2 pi^D / (D-1)!) + ln 1F1(1; D; kappa) in closed form.
#pragma once
#include "cacgmm_em.hpp"

namespace pbbss {

struct WatsonArgs {
  EmArgs em;                 // y, B, T, gamma0, saliency, iterations, weight_mode, layout, outputs
  const double* in_mode;     // c128 (B,K,D) or null
  const double* in_conc;     // (B,K)
  const double* spline_t;    // knots (n_coef + 3)
  const double* spline_c;    // coefficients (n_coef)
  int n_coef;
  double ev_min, ev_max;     // interpolation range of the eigenvalue (outside: fill values)
  double max_concentration;
  double* out_mode;          // c128 (B,K,D)
  double* out_conc;          // (B,K)
};

// ln n!, n <= 8 (the sensor counts of the fused kernels)
__device__ __forceinline__ constexpr double ln_factorial(int n) {
  constexpr double t[9] = {0.0, 0.0, 0.6931471805599453, 1.791759469228055, 3.1780538303479458,
                           4.787491742782046, 6.579251212010101, 8.525161361065415,
                           10.60460290274525};
  return t[n];
}

// ln 1F1(1; D; kappa) = ln sum_m kappa^m / (D)_m.  kappa < 8: the series itself, 48 terms with
// compile-time reciprocals (term 48 is < 1e-19 of the sum for D >= 2; the round-1 loop divided
// per term and ran to 100 terms at kappa = 25: a 7 us serial chain inside every factorisation).
// Otherwise the closed form
//   1F1(1; D; k) = (D-1)! k^-(D-1) (e^k - sum_{r<=D-2} k^r / r!),
// whose subtraction keeps >= 14 digits from kappa = 3 on (checked against mpmath, D = 2..8).
template <int D>
__device__ __forceinline__ double log_hyp1f1_1_D(double kappa) {
  static_assert(D >= 2 && D <= 9, "ln_factorial table");
  if (kappa < 8.0) {
    double s = 1.0, term = 1.0;
    static_for<1, 49>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      term *= kappa * (1.0 / (double)(D + m - 1));
      s += term;
    });
    return log_pos(s);  // s >= 1
  }
  double ssum = 1.0, term = 1.0;
  static_for<1, D - 1>([&](auto rc) {
    constexpr int r = decltype(rc)::value;
    term *= kappa * (1.0 / (double)r);
    ssum += term;
  });
  // kappa >= 8: e^-kappa ssum <= 0.32 for D <= 8 (D = 8, kappa = 8: 934 e^-8)
  return ln_factorial(D - 1) - (double)(D - 1) * log_pos(kappa) + kappa +
         log1p_small(-exp_nonpos(-kappa) * ssum);
}

// ln c(kappa) as complex_watson.py:157-168
template <int D>
__device__ __forceinline__ double watson_log_norm(double kappa) {
  return 0.6931471805599453 + (double)D * 1.1447298858494002 /* ln pi */ - ln_factorial(D - 1) +
         log_hyp1f1_1_D<D>(kappa);
}

constexpr int kWatsonKnotTable = 64;  // every S-th knot of the spline, staged in LDS

// A 64-knot window of the spline around the interval the class's eigenvalue fell into last time,
// one knot and one coefficient per lane, kept in the registers of the wave that owns the class.
// Between two EM iterations the eigenvalue moves by a fraction of a knot spacing, so the next
// evaluation finds its interval by a ballot over the window and fetches the seven operands of
// the de Boor recursion with v_readlane -- no global load on the factorisation's serial path
// (the two dependent loads of the search below were ~1.5 us of its 6.4 us; round 4).
struct SplineWin {
  int base = -1;  // knot index held by lane 0; < 0: empty
  double t = 0.0, c = 0.0;
};

// scipy.interpolate.interp1d(kind='quadratic', bounds_error=False, fill_value=(0, max))
// == BSpline(t, c, k=2) evaluated with de Boor inside [ev_min, ev_max].  Called by a whole
// wavefront with a wave-uniform `ev`: the knot interval comes from the window `win` when it
// covers it, else from a two-level 64-way search (level 1 = `knot1`, every S-th knot, in LDS;
// level 2 = one global load per lane) instead of a binary search (10 dependent global loads for
// the reference's 1000 markers).  Same operands, same arithmetic on both routes.
__device__ __forceinline__ double watson_concentration(const WatsonArgs& a, const double* knot1,
                                                       double ev, int lane, SplineWin& win) {
  if (!(ev >= a.ev_min)) return 0.0;  // also NaN -> 0 like fill_value below the range
  if (ev > a.ev_max) return a.max_concentration;
  constexpr int k = 2;
  const int n = a.n_coef;
  // largest i in [k, n-1] with t[i] <= ev (the knots ascend: the votes form a prefix)
  int i = -1, m = -1;
  if (win.base >= 0) {
    const int idx = win.base + lane;
    const unsigned long long v = __ballot(idx >= k && idx <= n - 1 && win.t <= ev);
    if (v != 0ull) {
      const int top = 63 - __builtin_clzll(v);
      // lanes top-2 .. top+2 hold every operand; lane top+1 did not vote, so t[i+1] > ev (or
      // i = n-1): i is the answer over the whole knot vector
      if (top >= 2 && top <= 61) {
        m = top;
        i = win.base + top;
      }
    }
  }
  double tk[4], ck[3];  // t[i-1 .. i+2], c[i-2 .. i]
  if (m >= 0) {
    m = __builtin_amdgcn_readfirstlane(m);
#pragma unroll
    for (int x = 0; x < 4; ++x) tk[x] = lane_bcast_const(win.t, m - 1 + x);
#pragma unroll
    for (int x = 0; x < 3; ++x) ck[x] = lane_bcast_const(win.c, m - 2 + x);
  } else {
    const int S = (n - k + kWatsonKnotTable - 1) / kWatsonKnotTable;
    if (S <= kWave) {
      const unsigned long long v1 = __ballot(k + lane * S < n && knot1[lane] <= ev);
      const int base = k + max(__popcll(v1) - 1, 0) * S;
      const bool in2 = lane < S && base + lane < n;
      const double t2 = a.spline_t[in2 ? base + lane : base];
      const unsigned long long v2 = __ballot(in2 && t2 <= ev);
      i = base + max(__popcll(v2) - 1, 0);
    } else {
      int lo = k, hi = n - 1;
      while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (a.spline_t[mid] <= ev) lo = mid; else hi = mid - 1;
      }
      i = lo;
    }
    i = __builtin_amdgcn_readfirstlane(i);
#pragma unroll
    for (int x = 0; x < 4; ++x) tk[x] = a.spline_t[i - 1 + x];
#pragma unroll
    for (int x = 0; x < 3; ++x) ck[x] = a.spline_c[i - 2 + x];
  }
  // (re-)centre the window when the interval is unknown to it or has drifted towards an edge;
  // the loads complete behind the phases that follow
  if (m < 12 || m > 51) {
    const int nb = min(max(i - 32, 0), max(n + 3 - kWave, 0));
    win.base = nb;
    win.t = a.spline_t[min(nb + lane, n + 2)];
    win.c = a.spline_c[min(nb + lane, n - 1)];
  }
  // de Boor, k = 2: d[j] = c[j + i - 2]; tl = t[j + i - 2], tr = t[j + 1 + i - r]
  double d[k + 1] = {ck[0], ck[1], ck[2]};
#pragma unroll
  for (int r = 1; r <= k; ++r) {
#pragma unroll
    for (int j = k; j >= r; --j) {
      const double tl = tk[j - 1], tr = tk[j + 2 - r];  // indices relative to i - 1
      double alpha = (ev - tl) * fast_rcp(tr - tl);  // distinct knots: a finite normal difference
      d[j] = (1.0 - alpha) * d[j - 1] + alpha * d[j];
    }
  }
  return d[k];
}

template <int D, int K, typename YS, bool SPILL>
struct WatsonKernel {
  using Base = EmKernel<D, K, YS, SPILL>;
  using Lds = typename Base::Lds;
  static constexpr int NA = Base::NA;
  // L.detm[k] holds kappa_k, L.rdet[k] holds ln c(kappa_k)

  // tf: first frame of this workgroup's window (split groups); the (B, K, T) / (B, T) arrays are
  // indexed with the whole problem's row stride
  // w_kt: frame-varying weights (K, T) of the problem's group (weight_constant_axis (-3,)) or null
  // (the per-class weights in LDS); pub_kt: !FINAL -- the masked affiliations (K, T) of this
  // problem for the group's reduction (WatsonShared)
  // NT: threads that share the frames of the bin (kEmThreads; 2 kEmThreads in WatsonWide, where
  // `tid` / `wave` run over both halves of the workgroup); red_out: [NT / 64][K] class sums of the
  // waves (null: L.red)
  template <bool FINAL, int NT = kEmThreads>
  static __device__ void phase_e(const WatsonArgs& wa, const Lds& L, int64_t b, int tid, int wave,
                                 int lane, int tf = 0, const double* w_kt = nullptr,
                                 double* pub_kt = nullptr, double* red_out = nullptr) {
    tid = opaque(tid);
    lane = opaque(lane);
    const EmArgs& a = wa.em;
    const int TS = Base::t_stride(a);
    double s[K];
#pragma unroll
    for (int k = 0; k < K; ++k) s[k] = 0.0;
    for (int t0 = 0; t0 < a.T; t0 += NT) {
      // the wave's 64 frames are one chunk of the frame arrays (cacgmm_em.hpp: Lds); beyond the
      // padded frame count there is nothing (wave-uniform exit: later passes lie further out)
      if (t0 + wave * kWave >= Base::padded_frames(a.T)) break;
      const int tt = t0 + tid;         // frame in LDS: padding frames read y = 0, 1/|y|^2 = 0
      const bool ok = tt < a.T;
      const int t = ok ? tt : a.T - 1;  // clamped for the HBM side arrays
      double re[D], im[D], q[K];
      Base::load_frame(L, tt, re, im);
      mode_forms(L, lane, re, im, q);  // |m_k^H y|^2
      const double inv = L.inv_n2[tt];
      double lp[K], mx = -1.79e308;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        lp[k] = L.detm[k] * (q[k] * inv) - L.rdet[k];  // complex_watson.py:83-86
        mx = fmax(mx, lp[k]);
      }
      double g[K], den = 0.0;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const double w = w_kt ? __hip_atomic_load(w_kt + (size_t)k * TS + tf + t, __ATOMIC_RELAXED,
                                                  __HIP_MEMORY_SCOPE_AGENT)
                              : L.wgt[k];
        g[k] = exp_nonpos(lp[k] - mx) * w;  // mixture_model_utils.py:32-37
        den += g[k];
      }
      // one Newton-refined reciprocal (~1 ulp) instead of an IEEE division; a non-finite class
      // sum stays visible (np.maximum keeps the NaN that v_max drops: den - den is 0 or NaN)
      const double rden = fast_rcp(fmax(den, kTiny)) + (den - den);
      const double sal = (!FINAL && a.saliency) ? a.saliency[(size_t)b * TS + tf + t] : 1.0;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        double gam = g[k] * rden;
        if constexpr (FINAL) {
          if (ok) {
            size_t idx = ((size_t)b * K + k) * TS + tf + t;
            if (a.out_aff) a.out_aff[idx] = gam;
            if (a.out_logpdf) a.out_logpdf[idx] = lp[k];
          }
        } else {
          double gs = ok ? gam * sal : 0.0;
          if (pub_kt && ok)
            __hip_atomic_store(pub_kt + (size_t)k * TS + tf + t, gs, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
          // complex_watson.py:306-309; unmasked: a padding frame stores 0 (gs = 0, inv = 0), as
          // the unmasked M sweep needs
          L.wbuf[Base::woff(k, tt)] = gs * inv;
          s[k] += gs;
        }
      }
    }
    if constexpr (!FINAL) wave_class_sums<K>(s, lane, (red_out ? red_out : L.red) + wave * K);
  }

  // affiliation initialisation -> M-step weights (cwmm.py:162-163 with saliency)
  static __device__ void phase_init_gamma(const EmArgs& a, const Lds& L, int64_t b, int tid,
                                          int wave, int lane, int tf = 0) {
    const int TS = Base::t_stride(a);
    double s[K];
#pragma unroll
    for (int k = 0; k < K; ++k) s[k] = 0.0;
    for (int t = tid; t < a.T; t += kEmThreads) {
      double sal = a.saliency ? a.saliency[(size_t)b * TS + tf + t] : 1.0;
      double inv = L.inv_n2[t];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const size_t idx = ((size_t)b * K + k) * TS + tf + t;
        double g = a.gamma0[idx] * sal;
        L.wbuf[Base::woff(k, t)] = g * inv;
        if (a.gaff && a.weight_mode == PBBSS_WEIGHT_SHARED_KT)  // parity 0: iteration 0
          __hip_atomic_store(a.gaff + idx, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s[k] += g;
      }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      double tot = wave_sum(s[k]);
      if (lane == 0) L.red[wave * K + k] = tot;
    }
  }

  // mode vector on lanes (i, 0) -> LDS; kappa, ln c
  static __device__ __forceinline__ void set_class(const Lds& L, int k, int lane, LaneIJ c,
                                                   double mre_i, double mim_i, double kappa) {
    // interleaved (Re m_0, Im m_0, Re m_1, ...) for the E phase's DPP operand register: parked in
    // this class's slice of the covariance-sum array, which is free between the factorisation
    // (its last reader) and the next M phase (which rewrites all of it); 2 D <= D^2 doubles
    if (c.j == 0 && c.i < D) {
      L.cpack[k * NA + 2 * c.i] = mre_i;
      L.cpack[k * NA + 2 * c.i + 1] = mim_i;
    }
    if (lane == 0) {
      L.detm[k] = kappa;
      L.rdet[k] = watson_log_norm<D>(kappa);
    }
  }

  // |m_k^H y|^2 for all classes.  The rank-one structure is used directly -- D complex
  // multiply-adds per class instead of the D^2 real ones of <m m^H, P> plus the outer product P
  // -- with the 2 D <= 16 components of m_k as DPP operands of ONE register per class
  // (pbbss_dev.hpp: fmac_row_bcast; lane l holds component l & 15 of the interleaved mode).
  static __device__ __forceinline__ void mode_forms(const Lds& L, int lane, const double (&re)[D],
                                                    const double (&im)[D], double (&q)[K]) {
    static_assert(2 * D <= 16, "one DPP operand register per class");
    const int comp = ((lane & 15) < 2 * D) ? (lane & 15) : 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const double m = L.cpack[k * NA + comp];
      double ur = 0.0, ui = 0.0, vi = 0.0;  // conj(m) y = (ur) + i (ui - vi)
      static_for<0, D>([&](auto dc) {
        constexpr int d = dc;
        fmac_row_bcast<2 * d>(ur, m, re[d]);      // Re m_d Re y_d
        fmac_row_bcast<2 * d + 1>(ur, m, im[d]);  // Im m_d Im y_d
        fmac_row_bcast<2 * d>(ui, m, im[d]);      // Re m_d Im y_d
        fmac_row_bcast<2 * d + 1>(vi, m, re[d]);  // Im m_d Re y_d
      });
      const double w = ui - vi;
      q[k] = fma(ur, ur, w * w);
    }
  }

  // (pvre, pvim): MODE of this class from the previous EM iteration (component i on the lanes of
  // row i), kept in the registers of the wave that owns the class.  The M-step keeps ONE
  // eigenpair (complex_watson.py:300-315, utils.get_pca): from the previous mode the dominant
  // eigenpair comes out of a few rounds of shifted inverse iteration on positive definite
  // systems (wave_la.hpp: wave_dominant_eigenpair) instead of a full Jacobi decomposition
  // (8.5-14.5 us of the 44 us iteration at D = 8; profiles/r03_l_api_sweep.txt).  The first
  // iteration, a start vector that tracked a non-dominant eigenpair (certified by the sign of the
  // pivots) and a residual that does not reach rounding level fall back to the cyclic Jacobi.
  // The eigenvector phase is free and cancels in m m^H.  `warm` is false on the first iteration.
  static __device__ void factor_class(const WatsonArgs& wa, const Lds& L, const double* knot1,
                                      const uint32_t* jtab, int64_t b, int k, int lane,
                                      bool last, bool warm, double& pvre, double& pvim,
                                      SplineWin& win) {
    lane = opaque(lane);
    const EmArgs& a = wa.em;
    // top issue priority for the serial chain (cacgmm_em.hpp: phase_m / factor_class)
    if (a.split_groups == 0) __builtin_amdgcn_s_setprio(3);
    const LaneIJ c = lane_ij(lane);
    const bool valid = c.i < D && c.j < D;
    // covariance entry of this lane first: its LDS round trip overlaps the class sums below
    double are = 0.0, aim = 0.0;
    if (valid) Base::cov_entry(L, k, c.i, c.j, are, aim);
    double S = 0.0, tot = 0.0;
#pragma unroll
    for (int kk = 0; kk < K; ++kk) {
      double sk = 0.0;
#pragma unroll
      for (int w = 0; w < kEmWaves; ++w) sk += L.red[w * K + kk];
      tot += fabs(sk);
      S = (kk == k) ? sk : S;
    }
    if (valid) {
      // complex_watson.py:311 (no floor in the reference).  A normal S takes the Newton-refined
      // reciprocal (~1 ulp; the IEEE division was ~30 dependent instructions at the head of the
      // class's serial chain), anything else -- zero, denormal, non-finite -- the division
      const double scale = (fabs(S) > 1e-300 && fabs(S) < 1e300) ? fast_rcp(S) : 1.0 / S;
      are *= scale;
      aim *= scale;
    }
    int st = 0;
    if (__ballot(!(isfinite(are) && isfinite(aim))) != 0ull) st |= PBBSS_ST_NONFINITE;
    double lmax = 0.0, mre_i = 0.0, mim_i = 0.0;
    bool fast = false;
    if (warm && !(st & PBBSS_ST_NONFINITE) && !a.force_eig) {
      double xr = pvre, xi = pvim, lam = 0.0;
      int rounds = 0;
      fast = wave_dominant_eigenpair<D>(are, aim, c, xr, xi, lam, rounds);
      fast = __builtin_amdgcn_readfirstlane((int)fast) != 0;
      if (fast) {
        lmax = lam;
        mre_i = xr;
        mim_i = xi;
      }
    }
    if (!fast) {
      double vre, vim;
      const int sweeps = wave_jacobi_heev_tab<D>(are, aim, c, vre, vim, jtab, lane);
      if (sweeps < 0) st |= PBBSS_ST_EIG_NOCONV;
      double lam = lane_get(are, ij_lane(c.j, c.j));
      int rank = wave_sort_rank<D>(lam, c);
      int col = 0;
#pragma unroll
      for (int m = 0; m < D; ++m) {
        int rm = lane_get(rank, ij_lane(0, m));
        double lm = lane_get(lam, ij_lane(0, m));
        if (rm == D - 1) {
          col = m;
          lmax = lm;
        }
      }
      // mode components for this lane's row index
      mre_i = lane_get(vre, ij_lane(c.i, col));
      mim_i = lane_get(vim, ij_lane(c.i, col));
    }
    pvre = (c.i < D) ? mre_i : 0.0;
    pvim = (c.i < D) ? mim_i : 0.0;
    const double kappa = watson_concentration(wa, knot1, lmax, lane, win);
    set_class(L, k, lane, c, mre_i, mim_i, kappa);
    if (last) {
      if (c.j == 0 && c.i < D && wa.out_mode) {
        double* o = wa.out_mode + (((size_t)b * K + k) * D + c.i) * 2;
        o[0] = mre_i;
        o[1] = mim_i;
      }
      if (lane == 0 && wa.out_conc) wa.out_conc[(size_t)b * K + k] = kappa;
    }
    if (lane == 0) {
      L.status[k] |= st;
      // the new mixture weight: off the head of the serial chain (one lane's division used to sit
      // in front of the eigenpair), nothing in this phase reads it
      if (a.weight_mode < PBBSS_WEIGHT_SHARED_K) {  // shared weights: WatsonShared
        // the reference always passes a saliency (ones by default, cwmm.py:134-135):
        // L1-normalised weighted sums (mixture_model_utils.py:192-201)
        L.wgt[k] = (a.weight_mode == PBBSS_WEIGHT_UNIFORM) ? 1.0 / K
                                                           : S / ((tot == 0.0) ? 1e-10 : tot);
      }
    }
    if (a.split_groups == 0) __builtin_amdgcn_s_setprio(0);
  }

  static __device__ void prep_from_model(const WatsonArgs& wa, const Lds& L, int64_t b, int k,
                                         int lane) {
    const EmArgs& a = wa.em;
    const LaneIJ c = lane_ij(lane);
    double mre_i = 0, mim_i = 0;
    if (c.i < D) {
      mre_i = wa.in_mode[(((size_t)b * K + k) * D + c.i) * 2];
      mim_i = wa.in_mode[(((size_t)b * K + k) * D + c.i) * 2 + 1];
    }
    set_class(L, k, lane, c, mre_i, mim_i, wa.in_conc[(size_t)b * K + k]);
    if (lane == 0) L.wgt[k] = a.in_weight ? a.in_weight[b * a.wb + k * a.wk] : 1.0 / K;
  }

  static __host__ __device__ size_t lds_bytes(int T) {
    return Base::lds_bytes(T) + kWatsonKnotTable * sizeof(double) +
           jacobi_table_dwords<D>() * sizeof(uint32_t);
  }

  static __device__ void run(const WatsonArgs& wa, char* smem) {
    const EmArgs& a = wa.em;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const Lds L = Base::carve(
        smem, a.T, SPILL ? a.scratch + (size_t)blockIdx.x * a.scratch_stride : nullptr);
    double* knot1 = reinterpret_cast<double*>(smem + Base::lds_bytes(a.T));
    uint32_t* jtab = reinterpret_cast<uint32_t*>(knot1 + kWatsonKnotTable);
    if (wave == 0) jacobi_table_build<D>(jtab, lane);  // visible after the loop's first barrier
    if (tid < kWatsonKnotTable && wa.spline_t && wa.n_coef > 2) {  // predict carries no spline
      const int S = (wa.n_coef - 2 + kWatsonKnotTable - 1) / kWatsonKnotTable;
      knot1[tid] = wa.spline_t[min(2 + tid * S, wa.n_coef - 1)];
    }
    for (int64_t b = blockIdx.x; b < a.B; b += gridDim.x) {
      __syncthreads();
      if (tid < K) L.status[tid] = 0;
      if (tid == 0) *L.flags = 0;
      __syncthreads();
      Base::phase_load(a, L, b, tid);
      __syncthreads();
      const bool model_in = (a.gamma0 == nullptr);
      if (model_in) {
        for (int k = wave; k < K; k += kEmWaves) prep_from_model(wa, L, b, k, lane);
      } else {
        phase_init_gamma(a, L, b, tid, wave, lane);
      }
      __syncthreads();
      double pvre = 0.0, pvim = 0.0;  // previous eigenvectors of class `wave` (K <= 4 <= waves)
      SplineWin win;                  // this class's window of the concentration spline
      for (int it = 0; it < a.iterations; ++it) {
#ifdef PBBSS_CW_DBG
        const bool dbg_on = it > 2 && it < a.iterations - 1;
#define PBBSS_CW_SKIP(bit) (dbg_on && (PBBSS_CW_DBG & (bit)))
#else
#define PBBSS_CW_SKIP(bit) false
#endif
        if ((it > 0 || model_in) && !PBBSS_CW_SKIP(4)) {
          phase_e<false>(wa, L, b, tid, wave, lane);
          __syncthreads();
        }
        if (!PBBSS_CW_SKIP(2)) {
        switch (wave) {
          case 0: Base::template phase_m<0>(a, L, lane); break;
          case 1: Base::template phase_m<1>(a, L, lane); break;
          case 2: Base::template phase_m<2>(a, L, lane); break;
          default: Base::template phase_m<3>(a, L, lane); break;
        }
        }
        __syncthreads();
        const bool last = (it == a.iterations - 1);
        static_assert(K <= kEmWaves, "one class per wave: the warm start lives in its registers");
        if (wave < K && !PBBSS_CW_SKIP(1))
          factor_class(wa, L, knot1, jtab, b, wave, lane, last, it > 0, pvre, pvim, win);
        __syncthreads();
      }
      if (tid < K) {
        if (a.out_weight && a.iterations > 0) a.out_weight[(size_t)b * K + tid] = L.wgt[tid];
        if (a.out_status) a.out_status[(size_t)b * K + tid] = L.status[tid];
      }
      if (a.final_predict) phase_e<true>(wa, L, b, tid, wave, lane);
    }
  }
};

// EIGHT wavefronts per bin (round 6).  BASELINE configs[3] has 257 bins for 256 compute units: one
// workgroup per CU, one wavefront per SIMD, and every phase of the four-wave kernel runs at the
// latency of its dependent instruction chains (s_memtime per phase, profiles/r06_d_watson_phases.txt:
// 5.8 cycles per issued instruction in E and M).  Here the frames of a bin are shared by two
// groups of four waves -- two wavefronts per SIMD that interleave:
//   E   chunk c of 64 frames belongs to wave c mod 8 (phase_e<.., 2 kEmThreads>)
//   M   group g accumulates its half of the chunks for the same four entry sets (phase_m with a
//       chunk range); group 1 writes its packed sums to a second array, a 108-element add merges
//   F   unchanged: wave k < K factors class k from the merged sums.
// One more workgroup barrier per iteration (the merge).  Used for the main launch when every
// bin has a compute unit of its own (cw_inst.hip); remainder bins stay with the split groups,
// spilled frames and shared weights with the four-wave kernels.
template <int D, int K, typename YS>
struct WatsonWide {
  using W = WatsonKernel<D, K, YS, false>;
  using Base = typename W::Base;
  using Lds = typename W::Lds;
  static constexpr int NT = 2 * kEmThreads;
  static constexpr int NA = Base::NA;

  // Base arrays | knot table | Jacobi table | second packed-sum array (a whole small block: the
  // write-back table addresses its padding sink relative to the array) | class sums of 8 waves
  static __host__ __device__ size_t lds_bytes(int T) {
    return W::lds_bytes(T) + ((Base::small_bytes() + 15) & ~(size_t)15) + 2 * kEmWaves * K * sizeof(double);
  }

  static __device__ void run(const WatsonArgs& wa, char* smem) {
    const EmArgs& a = wa.em;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int grp = wave >> 2, wv = wave & 3;
    const Lds L = Base::carve(smem, a.T, nullptr);
    double* knot1 = reinterpret_cast<double*>(smem + Base::lds_bytes(a.T));
    uint32_t* jtab = reinterpret_cast<uint32_t*>(knot1 + kWatsonKnotTable);
    char* extra = smem + W::lds_bytes(a.T);
    double* cpack2 = reinterpret_cast<double*>(extra);
    double* red8 = reinterpret_cast<double*>(extra + ((Base::small_bytes() + 15) & ~(size_t)15));
    if (wave == 0) jacobi_table_build<D>(jtab, lane);  // visible after the loop's first barrier
    if (tid < kWatsonKnotTable && wa.spline_t && wa.n_coef > 2) {  // predict carries no spline
      const int S = (wa.n_coef - 2 + kWatsonKnotTable - 1) / kWatsonKnotTable;
      knot1[tid] = wa.spline_t[min(2 + tid * S, wa.n_coef - 1)];
    }
    const int nchunk = Base::padded_frames(a.T) >> 6;
    const int csplit = (nchunk + 1) >> 1;  // group 0: chunks [0, csplit), group 1: the rest (>= 1)
    const int64_t b = blockIdx.x;          // one workgroup per problem
    if (tid < K) L.status[tid] = 0;
    if (tid == 0) *L.flags = 0;
    __syncthreads();
    if (grp == 0) Base::phase_load(a, L, b, tid);
    __syncthreads();
    const bool model_in = (a.gamma0 == nullptr);
    if (grp == 0) {
      if (model_in) {
        for (int k = wave; k < K; k += kEmWaves) W::prep_from_model(wa, L, b, k, lane);
      } else {
        W::phase_init_gamma(a, L, b, tid, wave, lane);  // class sums -> L.red
      }
    }
    __syncthreads();
    double pvre = 0.0, pvim = 0.0;  // previous eigenvectors of class `wave` (waves 0 .. K-1)
    SplineWin win;
    for (int it = 0; it < a.iterations; ++it) {
      const bool e_ran = (it > 0 || model_in);
      if (e_ran) {
        W::template phase_e<false, NT>(wa, L, b, tid, wave, lane, 0, nullptr, nullptr, red8);
        __syncthreads();
      }
      switch (wv) {
        case 0: Base::template phase_m<0>(a, L, lane, grp ? csplit : 0, grp ? nchunk : csplit, grp ? cpack2 : nullptr); break;
        case 1: Base::template phase_m<1>(a, L, lane, grp ? csplit : 0, grp ? nchunk : csplit, grp ? cpack2 : nullptr); break;
        case 2: Base::template phase_m<2>(a, L, lane, grp ? csplit : 0, grp ? nchunk : csplit, grp ? cpack2 : nullptr); break;
        default: Base::template phase_m<3>(a, L, lane, grp ? csplit : 0, grp ? nchunk : csplit, grp ? cpack2 : nullptr); break;
      }
      __syncthreads();
      // merge: packed sums of the two groups, class sums of the eight waves (a wave beyond the
      // last chunk wrote zeros)
      for (int i = tid; i < K * NA; i += NT) L.cpack[i] += cpack2[i];
      if (e_ran && tid < kEmWaves * K) L.red[tid] = red8[tid] + red8[tid + kEmWaves * K];
      __syncthreads();
      const bool last = (it == a.iterations - 1);
      if (wave < K) W::factor_class(wa, L, knot1, jtab, b, wave, lane, last, it > 0, pvre, pvim, win);
      __syncthreads();
    }
    if (tid < K) {
      if (a.out_weight && a.iterations > 0) a.out_weight[(size_t)b * K + tid] = L.wgt[tid];
      if (a.out_status) a.out_status[(size_t)b * K + tid] = L.status[tid];
    }
    if (a.final_predict) W::template phase_e<true, NT>(wa, L, b, tid, wave, lane);
  }
};

template <int D, int K, typename YS>
__global__ void __launch_bounds__(2 * kEmThreads, 2) cwmm_em_wide_kernel(WatsonArgs wa) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  WatsonWide<D, K, YS>::run(wa, smem);
}

template <int D, int K, typename YS, bool SPILL>
__global__ void __launch_bounds__(kEmThreads, em_waves_per_simd(K)) cwmm_em_kernel(WatsonArgs wa) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  WatsonKernel<D, K, YS, SPILL>::run(wa, smem);
}

// Mixture weights shared by a GROUP of problems -- weight_constant_axis = (-3, -1) of
// CWMMTrainer.fit (cwmm.py:217-240 with mixture_model_utils.py:184-201): the weights are averaged
// over the frequency bins of an utterance, the only quantity that couples the bins.  One
// cooperative launch, one workgroup per bin, all co-resident; per iteration every workgroup posts
// its K masked class sums and reads the group's sums back (EmKernel::shared_post /
// shared_acquire_k: the split-phase grid barrier of cacgmm_em_shared_kernel; the reference always
// passes a saliency -- ones by default --, hence the L1-normalised form).  Round 3 ran this
// option step by step: 0.26 instead of 0.09 ms per iteration.
template <int D, int K, typename YS>
struct WatsonShared {
  using W = WatsonKernel<D, K, YS, false>;
  using Base = typename W::Base;
  using Lds = typename W::Lds;

  static __device__ void run(const WatsonArgs& wa, char* smem) {
    const EmArgs& a = wa.em;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const Lds L = Base::carve(smem, a.T, nullptr);
    double* knot1 = reinterpret_cast<double*>(smem + Base::lds_bytes(a.T));
    uint32_t* jtab = reinterpret_cast<uint32_t*>(knot1 + kWatsonKnotTable);
    if (wave == 0) jacobi_table_build<D>(jtab, lane);
    if (tid < kWatsonKnotTable && wa.spline_t && wa.n_coef > 2) {
      const int S = (wa.n_coef - 2 + kWatsonKnotTable - 1) / kWatsonKnotTable;
      knot1[tid] = wa.spline_t[min(2 + tid * S, wa.n_coef - 1)];
    }
    const int64_t b = a.b_first + blockIdx.x;  // one workgroup per problem, all co-resident
    if (tid < K) L.status[tid] = 0;
    if (tid == 0) *L.flags = 0;
    __syncthreads();
    Base::phase_load(a, L, b, tid);
    __syncthreads();
    const bool kt = (a.weight_mode == PBBSS_WEIGHT_SHARED_KT);
    const int TS = Base::t_stride(a);
    W::phase_init_gamma(a, L, b, tid, wave, lane);  // the fit starts from affiliations
    if (kt) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    double pvre = 0.0, pvim = 0.0;
    SplineWin win;
    for (int it = 0; it < a.iterations; ++it) {
      if (it > 0) {
        if (kt) {  // (-3,): weights (K, T) of the group, the E-step publishes its masked affiliations
          double* pub = a.gaff + ((size_t)(it & 1) * a.B + b) * K * TS;
          const double* w_kt = Base::shared_acquire_kt(a, b, it - 1, tid);
          W::template phase_e<false>(wa, L, b, tid, wave, lane, 0, w_kt, pub);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {   // (-3, -1): weights (K) of the group -> L.wgt
          Base::shared_acquire_k(a, L, b, it - 1, tid, wave, lane);
          W::template phase_e<false>(wa, L, b, tid, wave, lane);
        }
        __syncthreads();
      }
      Base::shared_post(a, L, b, it, tid);
      switch (wave) {
        case 0: Base::template phase_m<0>(a, L, lane); break;
        case 1: Base::template phase_m<1>(a, L, lane); break;
        case 2: Base::template phase_m<2>(a, L, lane); break;
        default: Base::template phase_m<3>(a, L, lane); break;
      }
      __syncthreads();
      const bool last = (it == a.iterations - 1);
      if (wave < K) W::factor_class(wa, L, knot1, jtab, b, wave, lane, last, it > 0, pvre, pvim, win);
      __syncthreads();
      if (kt) Base::shared_reduce_kt(a, L, b, it, tid, wave, lane);
    }
    const int64_t grp = b / a.wgroup;
    const double* w_fin = nullptr;
    if (kt) {
      w_fin = Base::shared_acquire_kt(a, b, a.iterations - 1, tid);
      if (a.out_weight_shared && b == grp * a.wgroup) {
        for (int i = tid; i < K * TS; i += kEmThreads)
          a.out_weight_shared[(size_t)grp * K * TS + i] =
              __hip_atomic_load(w_fin + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      Base::shared_acquire_k(a, L, b, a.iterations - 1, tid, wave, lane);
      if (a.out_weight_shared && b == grp * a.wgroup && tid < K)
        a.out_weight_shared[(size_t)grp * K + tid] = L.wgt[tid];
    }
    if (tid < K && a.out_status) {
      int st = L.status[tid];
      if (Base::split_failed(a))
        st |= PBBSS_ST_EIG_NOCONV | PBBSS_ST_NONFINITE;  // a hand-off timed out: results are void
      a.out_status[(size_t)b * K + tid] = st;
    }
    if (a.final_predict) W::template phase_e<true>(wa, L, b, tid, wave, lane, 0, w_fin);
  }
};

template <int D, int K, typename YS>
__global__ void __launch_bounds__(kEmThreads, em_waves_per_simd(K)) cwmm_em_shared_kernel(WatsonArgs wa) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  WatsonShared<D, K, YS>::run(wa, smem);
}

// Remainder problems of 2^n + 1-bin utterances as split groups, the scheme of the cACGMM kernel
// (cacgmm_em.hpp: run_split): G member workgroups share one bin's frames (windows of
// `split_window` frames), run E and M on their window, exchange the packed covariance sums and
// class sums through L2 slabs (EmKernel::split_exchange -- the M-phase accumulators are the same
// arrays) and then EVERY member factors every class from the identical totals: mode, kappa and
// ln c need no hand-off, the factorisation of a class sits on one wavefront either way.  Only
// member 0 writes the model.
template <int D, int K, typename YS>
struct WatsonSplit {
  using W = WatsonKernel<D, K, YS, false>;
  using Base = typename W::Base;
  using Lds = typename W::Lds;

  static __host__ __device__ size_t lds_bytes(int window) { return W::lds_bytes(window); }

  static __device__ void run(const WatsonArgs& gwa, char* smem, int mblock, int nblocks) {
    const EmArgs& ga = gwa.em;
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
    WatsonArgs wa = gwa;  // this workgroup's window; only member 0 writes the model
    EmArgs& a = wa.em;
    a.T = min(ga.split_window, ga.T_total - tf);
    if (g != 0) {
      wa.out_mode = nullptr;
      wa.out_conc = nullptr;
    }
    const Lds L = Base::carve(smem, ga.split_window, nullptr);
    double* knot1 = reinterpret_cast<double*>(smem + Base::lds_bytes(ga.split_window));
    uint32_t* jtab = reinterpret_cast<uint32_t*>(knot1 + kWatsonKnotTable);
    if (wave == 0) jacobi_table_build<D>(jtab, lane);
    if (tid < kWatsonKnotTable && wa.spline_t && wa.n_coef > 2) {
      const int S = (wa.n_coef - 2 + kWatsonKnotTable - 1) / kWatsonKnotTable;
      knot1[tid] = wa.spline_t[min(2 + tid * S, wa.n_coef - 1)];
    }
    if (tid < K) {
      L.status[tid] = 0;
      // the members OR their bits into the status words at the end; member 0 zeroes them first
      // (a poison bit OR-ed in before this store is OR-ed in again by member 0 at its own end)
      if (g == 0 && a.out_status) a.out_status[(size_t)b * K + tid] = 0;
    }
    if (tid == 0) *L.flags = 0;
    __syncthreads();
    Base::phase_load(a, L, b, tid, tf);
    __syncthreads();
    const bool model_in = (a.gamma0 == nullptr);
    if (model_in) {
      for (int k = wave; k < K; k += kEmWaves) W::prep_from_model(wa, L, b, k, lane);
    } else {
      W::phase_init_gamma(a, L, b, tid, wave, lane, tf);
    }
    __syncthreads();
    double pvre = 0.0, pvim = 0.0;
    SplineWin win;
    for (int it = 0; it < a.iterations; ++it) {
      if (it > 0 || model_in) {
        // windows are <= 256 frames: waves that own no frame skip the phase
        if ((wave << 6) < a.T) {
          W::template phase_e<false>(wa, L, b, tid, wave, lane, tf);
        } else if (lane < K) {
          L.red[wave * K + lane] = 0.0;
        }
        __syncthreads();
      }
      switch (wave) {
        case 0: Base::template phase_m<0>(a, L, lane); break;
        case 1: Base::template phase_m<1>(a, L, lane); break;
        case 2: Base::template phase_m<2>(a, L, lane); break;
        default: Base::template phase_m<3>(a, L, lane); break;
      }
      __syncthreads();
      Base::split_exchange(a, L, prob, nprob, g, it, tid);
      const bool last = (it == a.iterations - 1);
      if (wave < K) W::factor_class(wa, L, knot1, jtab, b, wave, lane, last, it > 0, pvre, pvim, win);
      __syncthreads();
    }
    if (tid < K) {
      if (g == 0 && a.out_weight && a.iterations > 0) a.out_weight[(size_t)b * K + tid] = L.wgt[tid];
      const int bits = (g == 0 ? L.status[tid] : 0) |
                       (Base::split_failed(a) ? (PBBSS_ST_EIG_NOCONV | PBBSS_ST_NONFINITE) : 0);
      if (a.out_status && bits) atomicOr(a.out_status + (size_t)b * K + tid, bits);
    }
    if (a.final_predict) W::template phase_e<true>(wa, L, b, tid, wave, lane, tf);
    // the member that leaves last puts the problem's arrival counter back to zero
    if (tid == 0) {
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
};

template <int D, int K, typename YS>
__global__ void __launch_bounds__(kEmThreads, em_waves_per_simd(K)) cwmm_em_split_kernel(WatsonArgs wa) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  WatsonSplit<D, K, YS>::run(wa, smem, blockIdx.x, gridDim.x);
}

}  // namespace pbbss
