// Persistent cACGMM EM kernel for gfx950 (MI355X).
//
// One 256-thread workgroup (4 waves, one per SIMD) owns one independent problem
// b (a frequency bin of one utterance) for the WHOLE EM loop: the observation
// y[b] is read from HBM once, kept in LDS, and every iteration (E-step,
// M-step, factorisation) runs on-chip.  Reference loop being replaced:
// distribution/cacgmm.py:252-278 (fit), :73-95 (_predict), :315-345 (_m_step).
//
// Math (float64 throughout; see DESIGN.md "kernel math"):
//   P_t   = y_t y_t^H / |y_t|^2          Hermitian outer product, D^2 reals
//   q_kt  = <A_k, P_t>  with A_k = B_k^-1   (cacg.py:185-199; the reference's
//                                            einsum also forms B^-1 first)
//   gamma = softmax_k(-D ln q_kt - ln det B_k) * pi_k, clipped (mixture_model_utils.py:7-55)
//   C_k   = D * sum_t gamma_kt/q_kt P_t / sum_t gamma_kt   (cacg.py:316-327)
//   B_k   = eigen-floored C_k               (cacg.py:82-132)
// Inside the loop B_k^-1 and det B_k come from a Cholesky factorisation of C_k
// whenever a cheap bound proves no eigenvalue can reach the relative floor
// (posteriors are invariant to the scale of B_k, cacg.py:113 "The scale of
// the eigenvals does not matter"); otherwise, and always for the returned
// model, a Jacobi eigendecomposition applies the reference's normalisation
// and floor exactly.
//
// Phases per iteration (barrier separated):
//   E  frames split over the 4 waves, lane = frame: q, posteriors, and the
//      M-step weights w_kt = gamma_kt/q_kt/|y_t|^2 -> LDS
//   M  Hermitian entries split over the 4 waves, every wave sweeps all frames,
//      lane = frame: acc_k[e] += w_kt P_t[e]; wave butterfly; -> LDS
//   F  wave k factors C_k (lane = matrix entry): Cholesky / inverse or Jacobi
#pragma once
#include "pbbss.h"
#include "wave_la.hpp"
#include "embed_dev.hpp"

namespace pbbss {

struct EmArgs {
  const void* y;
  int64_t B;
  int T;
  // initialisation
  const double* gamma0;  // (B,K,T) or null
  const double* q0;      // (B,K,T) or null (=ones)
  const double* in_eigvec;  // c128 (B,K,D,D) interleaved, or null
  const double* in_eigval;  // (B,K,D)
  const double* in_weight;  // strided, see wb/wk/wt
  int64_t wb, wk, wt;
  const double* saliency;   // (B,T) or null
  const uint8_t* activity;        // (B,K,T) or null: source_activity_mask of the EM loop
  const uint8_t* final_activity;  // mask of the final predict (CACGMM.predict argument)
  // outputs (any may be null)
  double* out_eigvec;
  double* out_eigval;
  double* out_weight;
  int32_t* out_status;
  double* out_aff;
  double* out_q;
  double* out_logpdf;
  double* out_cov;  // c128 (B,K,D,D): covariance of the last M-step (after /denominator)
  // per-workgroup HBM scratch for the frame-sized arrays when they exceed LDS
  char* scratch;
  size_t scratch_stride;
  // optional phase cycle counters (only honoured by builds with -DPBBSS_PHASE_PROFILE)
  unsigned long long* prof;
  // ---- split-bin variant (SPLIT=true): G workgroups share the frames of one problem ----
  // a.T is then the frame count of ONE workgroup's window; global frame = t_first + t
  int T_total;        // frames of the whole problem (row stride of (B,K,T) arrays); 0 => a.T
  int split_groups;   // G
  int split_window;   // frames per workgroup window (multiple of 64)
  int64_t b_first;    // first problem handled by this launch
  double* xslab;      // [2][n_problems][G][slab_len] partial sums exchanged through L2
  unsigned* xcount;   // [n_problems] arrival counters (zeroed before the launch)
  int* xerror;        // [0] set to `xepoch` if a bounded spin of THIS launch ran out; [16] sticky copy
  int xepoch;         // launch stamp of the split protocol (0: the word is cleared by the launcher)
  int split_prio;     // s_setprio level of the split waves (0..3)
  // packed-FP32 kernel (cacgmm_em32.hpp): blocks >= main_grid are the member workgroups of the
  // remainder problems, in the same grid; 0: every block is a full workgroup
  int main_grid;
  unsigned lds_given;  // dynamic LDS bytes of the launch (checked by the debug build only)
  unsigned xbuf_given; // bytes of the exchange buffer behind xcount (debug build)
  unsigned spin_limit; // polls of a bounded inter-workgroup wait before it gives up (0: default;
                       // pbbss_set_spin_limit, a test knob that provokes REAL time-outs)
  // ---- weights shared across problems (run_shared: weight_mode PBBSS_WEIGHT_SHARED_*) ----
  int wgroup;          // problems (frequency bins) that share one set of mixture weights
  double* gsum;        // SHARED_K : [2][B][K]     masked class sums of every problem
  double* gaff;        // SHARED_KT: [2][B][K][T]  masked affiliations of every problem
  double* gw;          // SHARED_KT: [2][B / wgroup][K][T] reduced weights
  unsigned* gcount;    // [B / wgroup][2] arrival counters: posts, finished reducers
  double* out_weight_shared;  // (B / wgroup, K) or (B / wgroup, K, T)
  // options
  int iterations;
  int covariance_norm;
  int weight_mode;
  int layout;
  int final_predict;
  int force_eig;
  double aff_eps;
  double final_eps;
  double eig_floor;
};

// Extra inputs of the joint spatial+spectral E-step (gcacgmm.py:66-117, vmfcacgmm.py:57-97):
//   log p_k(t) = spatial_weight * cACG_log_pdf_{perm[k]}(t) + extra_logpdf[b,k,t]
// (extra_logpdf already carries the spectral weight).  perm == null: identity.
struct JointExtras {
  const double* extra_logpdf;  // (B,K,T)
  double spatial_weight;
  const int* perm;             // [K] in LDS: spatial class used by class slot k (inline PA)
  // Between the iterations of one fit the cACG model travels as the packed inverse
  // covariance A_k and det B_k ((B,K,D*D+2) float64) instead of (V, lambda): the M-step can
  // then take the Gauss-Jordan fast path, and only the last iteration pays for the
  // eigendecomposition the caller sees.
  const double* state_in;      // null: build A_k from a.in_eigvec / a.in_eigval
  double* state_out;           // null: do not export
  int emit_model;              // 1: exact eigen path, write a.out_eigvec / a.out_eigval
  // class weights of weight_constant_axis=(-1,) (gcacgmm.py:291-295): w[b,k] = sum_t masked
  // affiliation / sum over k, written by the workgroup that owns bin b (may alias a.in_weight)
  double* weight_fk_out;
  // Remainder problems of a launch (B = m * CUs + r): blocks >= main_grid are MEMBERS, G of
  // them share one of the r problems [a.b_first, a.b_first + r) by frame windows (run_joint_member);
  // 0: every block is a full workgroup.
  int main_grid;
  unsigned lds_given;  // dynamic LDS bytes of the launch (checked by the debug build only)
  unsigned xbuf_given; // bytes of the exchange buffer behind xcount (debug build)
};

// Spatial half of the ROTATED joint loop (round 4; pbbss_joint_fit).  The
// joint models read the embedding once per EM iteration: one sweep kernel (embed.hip:
// joint_sweep_kernel) evaluates the spectral log-pdf of a tile of embedding rows, combines it with
// the spatial quadratic forms Q of the same points, forms the posteriors and accumulates the
// spectral M-step sums from the SAME tile.  That moves the cACG E-step in front of the sweep:
//   S(it)  [this kernel, one workgroup per bin]  M-step weights from the posteriors G and the
//          quadratic forms Q of iteration it (phase_init_gamma: gamma0 = G, q0 = Q), covariance
//          sums, factorisation  ->  model(it)  ->  quadratic forms of model(it) -> Q (in place),
//          ln det B_k -> lndet
//   P(it+1) [sweep]  posteriors of model(it)  ->  G, spectral sums
// mode 0: E part only, model from a.in_eigvec / a.in_eigval (after the first M-step);
// mode 1: M + F + E;  mode 2: M + F with the exact eigen path, (V, lambda) emitted, no E (last).
struct JointMs {
  int mode;
  double* q_out;          // (B,K,T) quadratic forms max(|y^H B_k^-1 y|, tiny) of the new model
  double* lndet_out;      // (B,K)   ln det B_k of the new model
  double* weight_fk_out;  // (B,K) or null: class weights of weight_constant_axis=(-1,)
  unsigned lds_given;     // dynamic LDS bytes of the launch (debug build)
  // Grid: [0, main_grid) blocks over the bins (stride main_grid), then the HELPERS of the spectral
  // finalize (embed_dev.hpp: SpectralFin), run beside the bins of the same launch; fin.kind < 0
  // or fin.helpers == 0: none.  (Remainder bins as member blocks on frame windows -- the scheme of
  // run_joint_member -- were built and measured for this kernel: the last arriver's serial tail
  // (sum, factorisation, quadratic forms of every window) made the launch 8.5 us LONGER than
  // the third workgroup on one CU it was meant to avoid: 38.1 vs 29.5 us, profiles/r04_*.)
  int main_grid;
  SpectralFin fin;
};

// SPILL=false: observation, norms and M-step weights live in LDS (the fast path).
// SPILL=true : those three frame-sized arrays live in a per-workgroup HBM/L2
//              scratch slab (long utterances); the small matrices stay in LDS.
template <int D, int K, typename YS, bool SPILL = false>
struct EmKernel {
  static constexpr int DP = (D + 1) / 2;
  static constexpr int NOFF = D * (D - 1) / 2;
  static constexpr int NA = D * D;  // packed reals of one Hermitian matrix
  static constexpr int NDW = (D + kEmWaves - 1) / kEmWaves;     // diag entries per wave (max)
  static constexpr int NOW = (NOFF + kEmWaves - 1) / kEmWaves;  // off-diag pairs per wave (max)
#ifndef PBBSS_E_CHUNK
#define PBBSS_E_CHUNK 1
#endif
#ifndef PBBSS_E_PAIR
#define PBBSS_E_PAIR 0
#endif
#ifndef PBBSS_M_PREFETCH
#define PBBSS_M_PREFETCH 0
#endif
  static constexpr int kOperandChunk = PBBSS_E_CHUNK;  // pairs of A_k operands per prefetch stage of the E phase
  using YS4 = typename std::conditional<std::is_same<YS, float>::value, float4, double4>::type;
  using YS2 = typename std::conditional<std::is_same<YS, float>::value, float2, double2>::type;

  struct Lds {
    // Frame-sized arrays are kept in CHUNKS of 64 frames (one wavefront's worth), the planes of a
    // chunk back to back: inside a chunk every plane sits at a compile-time offset from the lane's
    // address, so a sweep over the frames advances ONE address register per array and the plane
    // offsets ride in the instructions' immediate fields (round 5: the [plane][Tp] layout cost
    // the M phase 13 VALU address instructions per 64 frames).  Tp is a multiple of 64; frames
    // [T, Tp) hold y = 0, 1/|y|^2 = 0, w = 0, so every sweep runs over whole chunks unmasked.
    YS* ybuf;        // [Tp/64][DP][64][4]  channel pairs (yoff)
    double* inv_n2;  // [Tp]
    double* wbuf;    // [Tp/64][K][64]      M-step weights (woff)
    double* cpack;   // [K][NA]       covariance sums, packed like apack: diag, then (Re, Im) per pair i<j
    double* apack;   // [K][NA]       A_k packed for the dot with P: diag, then (2Re, 2Im) per pair
    double* wgt;     // [K]           mixture weights
    double* detm;    // [K]           det B_k mantissa
    double* rdet;    // [K]           1 / detm
    double* red;     // [kEmWaves][K] cross-wave partial sums
    double* ssum;    // [K]           sum_t gamma_kt (saliency applied)
    int* dete;       // [K]           det B_k exponent
    int* status;     // [K]
    int* flags;      // [1]  bit0: the problem contains an all-zero frame
    unsigned short* wbtab;  // [kEmWaves][16][NACC/16] M-phase write-back: index into cpack per value
    double* sink;    // [1]  where the butterfly's padding values go
    int Tp;
  };

  static constexpr int kFC = 64;  // frames per chunk of the frame-sized arrays
  static __host__ __device__ __forceinline__ int padded_frames(int T) { return (T + kFC - 1) & ~(kFC - 1); }
  // element offsets into ybuf (in YS units) / wbuf of frame t
  static __host__ __device__ __forceinline__ int yoff(int dp, int t) {
    return ((((t >> 6) * DP + dp) << 6) + (t & 63)) * 4;
  }
  static __host__ __device__ __forceinline__ int woff(int k, int t) {
    return (((t >> 6) * K + k) << 6) + (t & 63);
  }
  static __host__ __device__ size_t frame_bytes(int T) {
    size_t Tp = (size_t)padded_frames(T);
    return (size_t)DP * Tp * 4 * sizeof(YS) + Tp * 8 + (size_t)K * Tp * 8;
  }
  static __host__ __device__ size_t small_bytes() {
    size_t n = 0;
    n += (size_t)K * NA * 8;        // cpack
    n += (size_t)K * NA * 8;        // apack
    n += (size_t)K * 8 * 4;         // wgt, detm, rdet, ssum
    n += (size_t)kEmWaves * K * 8;  // red
    n += (size_t)K * 4 * 2 + 16;    // dete, status, flags
    n += wbtab_bytes() + 8;         // wbtab, sink
    return n;
  }
  static __host__ __device__ size_t lds_bytes(int T) {
    size_t n = small_bytes() + (SPILL ? 0 : frame_bytes(T));
    return (n + 15) & ~(size_t)15;
  }
  static __host__ __device__ size_t scratch_bytes(int T) {
    return SPILL ? ((frame_bytes(T) + 255) & ~(size_t)255) : 0;
  }

  // the small (frame-independent) arrays alone, at p (the packed-FP32 kernel keeps its own
  // frame arrays and reuses the float64 factorisation / exchange code on these)
  static __device__ Lds carve_small(char* p, int Tp) {
    Lds L;
    L.Tp = Tp;
    L.ybuf = nullptr;
    L.inv_n2 = nullptr;
    L.wbuf = nullptr;
    carve_small_into(L, p);
    return L;
  }

  static __device__ Lds carve(char* base, int T, char* scratch = nullptr) {
    Lds L;
    L.Tp = padded_frames(T);
    char* p = base;
    char* f = SPILL ? scratch : base;  // frame-sized arrays
    L.ybuf = reinterpret_cast<YS*>(f);
    f += (size_t)DP * L.Tp * 4 * sizeof(YS);
    L.inv_n2 = reinterpret_cast<double*>(f);
    f += (size_t)L.Tp * 8;
    L.wbuf = reinterpret_cast<double*>(f);
    f += (size_t)K * L.Tp * 8;
    if (!SPILL) p = f;
    carve_small_into(L, p);
    fill_wbtab(L);
    return L;
  }

  static __device__ __forceinline__ void carve_small_into(Lds& L, char* p) {
    L.cpack = reinterpret_cast<double*>(p);
    p += (size_t)K * NA * 8;
    L.apack = reinterpret_cast<double*>(p);
    p += (size_t)K * NA * 8;
    L.wgt = reinterpret_cast<double*>(p);
    p += K * 8;
    L.detm = reinterpret_cast<double*>(p);
    p += K * 8;
    L.rdet = reinterpret_cast<double*>(p);
    p += K * 8;
    L.ssum = reinterpret_cast<double*>(p);
    p += K * 8;
    L.red = reinterpret_cast<double*>(p);
    p += kEmWaves * K * 8;
    L.dete = reinterpret_cast<int*>(p);
    p += K * 4;
    L.status = reinterpret_cast<int*>(p);
    p += K * 4;
    L.flags = reinterpret_cast<int*>(p);
    p += 16;
    L.wbtab = reinterpret_cast<unsigned short*>(p);
    p += wbtab_bytes();
    L.sink = reinterpret_cast<double*>(p);
  }

  // row stride of the (B,K,T)/(B,T) arrays and first global frame of this workgroup
  static __device__ __forceinline__ int t_stride(const EmArgs& a) {
    return a.T_total ? a.T_total : a.T;
  }

  // ---- observation frame t from LDS, widened to float64 -------------------
  static __device__ __forceinline__ void load_frame(const Lds& L, int t, double (&re)[D],
                                                    double (&im)[D]) {
    PBBSS_DEV_ASSERT(t >= 0 && t < L.Tp);
    static_for<0, DP>([&](auto dpc) {
      constexpr int dp = dpc;
      YS4 v = *reinterpret_cast<const YS4*>(L.ybuf + yoff(dp, t));
      re[2 * dp] = (double)v.x;
      im[2 * dp] = (double)v.y;
      if constexpr (2 * dp + 1 < D) {
        re[2 * dp + 1] = (double)v.z;
        im[2 * dp + 1] = (double)v.w;
      }
    });
  }

  // ---- phase L: HBM -> LDS, squared norms ---------------------------------
  static __device__ void phase_load(const EmArgs& a, const Lds& L, int64_t b, int tid,
                                    int tf = 0) {
    const int T = a.T;
    const int TS = t_stride(a);
    const YS2* yg = reinterpret_cast<const YS2*>(a.y);
    bool zero_seen = false;
    for (int t = tid; t < L.Tp; t += kEmThreads) {
      double n2 = 0.0;
      YS vr[2 * DP], vi[2 * DP];
#pragma unroll
      for (int d = 0; d < 2 * DP; ++d) {
        vr[d] = (YS)0;
        vi[d] = (YS)0;
      }
      if (t < T) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
          size_t idx = (a.layout == PBBSS_LAYOUT_TD) ? ((size_t)b * TS + tf + t) * D + d
                                                     : ((size_t)b * D + d) * TS + tf + t;
          YS2 v = yg[idx];
          vr[d] = v.x;
          vi[d] = v.y;
          n2 += (double)v.x * (double)v.x + (double)v.y * (double)v.y;
        }
      }
#pragma unroll
      for (int dp = 0; dp < DP; ++dp) {
        YS4 o;
        o.x = vr[2 * dp];
        o.y = vi[2 * dp];
        o.z = vr[2 * dp + 1];
        o.w = vi[2 * dp + 1];
        *reinterpret_cast<YS4*>(L.ybuf + yoff(dp, t)) = o;
      }
      double inv;
      if (a.layout == PBBSS_LAYOUT_TD) {
        inv = (n2 > 0.0) ? 1.0 / n2 : 0.0;  // unit-norm, zero frames stay zero (utils.py:251)
      } else {
        inv = 1.0;  // caller already normalised (as _predict / _fit receive it)
      }
      L.inv_n2[t] = (t < T) ? inv : 0.0;
      if (t < T && !(n2 > 0.0)) zero_seen = true;
      if (t >= T) {  // padding frames of the last chunk: weight 0 until an E phase rewrites them
#pragma unroll
        for (int k = 0; k < K; ++k) L.wbuf[woff(k, t)] = 0.0;
      }
    }
    // An all-zero frame has q floored at `tiny` whatever the scale of B_k
    // (cacg.py:185-199), so its posterior depends on the eigenvalue
    // normalisation: such problems must take the exact eigen path.
    if (zero_seen) atomicOr(L.flags, 1);
  }

  // ---- M-step weight of one frame/class  (cacg.py:310, :322) ---------------
  static __device__ __forceinline__ double mweight(double g_sal, double q, double inv_n2) {
    return g_sal / fmax(q, 10.0 * kTiny) * inv_n2;
  }

  // ---- phase I: weights from an affiliation initialisation -----------------
  static __device__ void phase_init_gamma(const EmArgs& a, const Lds& L, int64_t b, int tid,
                                          int wave, int lane, int tf = 0) {
    const int TS = t_stride(a);
    double s[K];
#pragma unroll
    for (int k = 0; k < K; ++k) s[k] = 0.0;
    for (int t = tid; t < a.T; t += kEmThreads) {
      double sal = a.saliency ? a.saliency[(size_t)b * TS + tf + t] : 1.0;
      double inv = L.inv_n2[t];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        size_t idx = ((size_t)b * K + k) * TS + tf + t;
        double g = a.gamma0[idx] * sal;
        double q = a.q0 ? a.q0[idx] : 1.0;
        L.wbuf[woff(k, t)] = mweight(g, q, inv);
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

  // ---- phase E -------------------------------------------------------------
  // FINAL=false: in-loop E-step -> wbuf + class sums.
  // FINAL=true : posteriors / quadratic form / log-pdf to HBM.
  // TW: mixture weights vary over frames (weight_constant_axis=-3 models,
  //     cacgmm.py:59) and are read with strides from HBM; otherwise the
  //     per-class weights in LDS are used.  (A runtime flag here makes hipcc
  //     unswitch the frame loop and spill ~300 VGPRs, hence a template.)
  // PAIR: frames are taken two per lane (512 per pass) while more than 256 remain -- every A_k
  //     operand fetched from LDS then feeds two frames -- and one per lane for the rest
  //     (T = 500: one paired pass; T <= 256: one single pass).
  // PERM: the joint E-step reads the spatial class of slot k through jx->perm (inline
  // permutation alignment); a compile-time flag, so that the default joint kernel does not
  // carry the selects' registers (the run-time test cost 224 bytes of scratch per lane)
  template <bool FINAL, bool TW, bool JOINT = false, bool PAIR = false, bool PERM = false>
  static __device__ void phase_e(const EmArgs& a, const Lds& L, int64_t b, int tid, int wave,
                                 int lane, double eps, int tf = 0,
                                 const JointExtras* jx = nullptr,
                                 const double* w_kt = nullptr,  // TW: weights (K, T) of this problem's group
                                 double* pub_kt = nullptr) {    // !FINAL: masked affiliation (K, T) out
    tid = opaque(tid);
    lane = opaque(lane);
    const int TS = t_stride(a);
    static_assert(!JOINT || !PAIR, "joint E-step is written for one frame per lane");
    double s[K];
#pragma unroll
    for (int k = 0; k < K; ++k) s[k] = 0.0;
    double jlogdet[K];  // ln det B_k of the joint (log-domain) E-step, once per phase
#pragma unroll
    for (int k = 0; k < K; ++k)
      jlogdet[k] = JOINT ? log(L.detm[k]) + (double)L.dete[k] * 0.6931471805599453 : 0.0;
    auto pass = [&](int t0, auto nfc) {
      constexpr int NF = decltype(nfc)::value;  // frames per lane in this pass
      // the wave's 64 frames of this pass are one chunk of the frame arrays: a chunk beyond the
      // padded frame count has nothing in it (wave-uniform: the whole wave leaves)
      if (t0 + wave * kWave >= padded_frames(a.T)) return;
      int tl[NF];  // frame in LDS (padding frames [T, Tp) read y = 0, 1/|y|^2 = 0)
      int tt[NF];  // frame clamped to [0, T) for the HBM side arrays
      bool ok[NF];
      bool inl[NF];  // the frame exists in LDS (always for f = 0: the chunk test above)
      double re[NF][D], im[NF][D], q[NF][K];
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        int t = t0 + f * kEmThreads + tid;
        ok[f] = t < a.T;
        inl[f] = (f == 0) || t < padded_frames(a.T);
        tt[f] = ok[f] ? t : (a.T - 1);
        tl[f] = (f == 0) ? t : min(t, padded_frames(a.T) - 1);
        load_frame(L, tl[f], re[f], im[f]);
#pragma unroll
        for (int k = 0; k < K; ++k) q[f][k] = 0.0;
      }
      // q_k = <A_k, P>: diagonal, then the strict upper triangle
      {
        // A_k as DPP operands: lane l keeps apack[k][16 g + (l & 15)] (one conflict-free
        // ds_read_b64 per 16 operands instead of one broadcast read per operand), and every
        // FMA of the dot product <A_k, P> picks its operand out of that register with
        // row_newbcast -- no LDS traffic inside the dot product, no operand staging.
        constexpr int NG = (NA + 15) / 16;
        double areg[K][NG];
#pragma unroll
        for (int k = 0; k < K; ++k) {
#pragma unroll
          for (int g = 0; g < NG; ++g) {
            const int e = 16 * g + (lane & 15);
            areg[k][g] = L.apack[k * NA + (e < NA ? e : NA - 1)];
          }
        }
        static_for<0, D>([&](auto ic) {
          constexpr int i = ic;
#pragma unroll
          for (int f = 0; f < NF; ++f) {
            const double dg = re[f][i] * re[f][i] + im[f][i] * im[f][i];
#pragma unroll
            for (int k = 0; k < K; ++k) fmac_row_bcast<i % 16>(q[f][k], areg[k][i / 16], dg);
          }
        });
        static_for<0, NOFF>([&](auto pc) {
          constexpr int p = pc;
          constexpr int i = tri_i<D>(p), j = tri_j<D>(p);
          constexpr int e0 = D + 2 * p, e1 = e0 + 1;
#pragma unroll
          for (int f = 0; f < NF; ++f) {
            const double pr = re[f][i] * re[f][j] + im[f][i] * im[f][j];   // Re y_i conj(y_j)
            const double pim = im[f][i] * re[f][j] - re[f][i] * im[f][j];  // Im y_i conj(y_j)
#pragma unroll
            for (int k = 0; k < K; ++k) {
              fmac_row_bcast<e0 % 16>(q[f][k], areg[k][e0 / 16], pr);
              fmac_row_bcast<e1 % 16>(q[f][k], areg[k][e1 / 16], pim);
            }
          }
        });
      }
      // per-class constants (fetched here, not at phase entry: keeping them live
      // across the operand loop costs spills)
      double detm[K], rdet[K], wgt[K];
      int dete[K];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        detm[k] = L.detm[k];
        rdet[k] = L.rdet[k];
        dete[k] = L.dete[k];
        wgt[k] = L.wgt[k];
      }
      if constexpr (JOINT) {
        // softmax of the weighted sum of spatial and spectral log-pdfs
        // (gcacgmm.py:108-115 -> mixture_model_utils.py:30-53)
        const int t = tt[0];
        const double inv = L.inv_n2[tl[0]];
        double g[K], den = 0.0;
        if (jx->spatial_weight == 1.0) {
          // spatial_weight = 1 (the default): exp(spatial log-pdf) = 1 / (det_k q_k^D) in the
          // mantissa / exponent form of the persistent kernel -- no logarithm; only the exponents
          // and the spectral log-pdf enter the max-shift (the mantissa factor lies in [1, 2^(D+1)]):
          //   g_k = rdet_k rm_k^D * exp((ex_k ln 2 + spectral_k) - max) * w_k
          double val[K], tk[K];
#pragma unroll
          for (int k = 0; k < K; ++k) {
            const double qq = fmax(fabs(q[0][k] * inv), kTiny);  // cacg.py:185-199
            q[0][k] = qq;
            int e;
            const double m = frexp(qq, &e);
            const double rm = fast_rcp(m);
            val[k] = rdet[k] * ipow<D>(rm);
            tk[k] = -(double)(e * D + dete[k]) * 0.6931471805599453;
          }
          double lt[K], vsel[K], mx = -1.79e308;
#pragma unroll
          for (int k = 0; k < K; ++k) {
            double sv = val[k], st = tk[k];
            if constexpr (PERM) {
              const int pk = jx->perm[k];
#pragma unroll
              for (int j = 0; j < K; ++j) {
                sv = (pk == j) ? val[j] : sv;
                st = (pk == j) ? tk[j] : st;
              }
            }
            vsel[k] = sv;
            lt[k] = st + jx->extra_logpdf[((size_t)b * K + k) * TS + tf + t];
            mx = fmax(mx, lt[k]);
          }
#pragma unroll
          for (int k = 0; k < K; ++k) {
            const double w = TW ? a.in_weight[b * a.wb + k * a.wk + (int64_t)(tf + t) * a.wt] : wgt[k];
            g[k] = vsel[k] * exp(lt[k] - mx) * w;
            den += g[k];
          }
        } else {
          double lps[K], lp[K], mx = -1.79e308;
#pragma unroll
          for (int k = 0; k < K; ++k) {
            double qq = fmax(fabs(q[0][k] * inv), kTiny);  // cacg.py:185-199
            q[0][k] = qq;
            lps[k] = -(double)D * log(qq) - jlogdet[k];
          }
#pragma unroll
          for (int k = 0; k < K; ++k) {
            double sp = lps[k];
            if constexpr (PERM) {
              const int pk = jx->perm[k];
#pragma unroll
              for (int j = 0; j < K; ++j) sp = (pk == j) ? lps[j] : sp;
            }
            lp[k] = jx->spatial_weight * sp + jx->extra_logpdf[((size_t)b * K + k) * TS + tf + t];
            mx = fmax(mx, lp[k]);
          }
#pragma unroll
          for (int k = 0; k < K; ++k) {
            double w = TW ? a.in_weight[b * a.wb + k * a.wk + (int64_t)(tf + t) * a.wt] : wgt[k];
            g[k] = exp(lp[k] - mx) * w;
            den += g[k];
          }
        }
        const double poison = den - den;  // 0, or NaN for a non-finite sum (see the non-joint branch)
        den = fmax(den, kTiny);
        const double rden = fast_rcp(den);  // one reciprocal instead of K divisions
        const double sal = (!FINAL && a.saliency) ? a.saliency[(size_t)b * TS + tf + t] : 1.0;
        const double invp = inv + poison;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          double gam = g[k] * rden;
          if (!FINAL || eps != 0.0) gam = fmin(fmax(gam, eps), 1.0 - eps);  // (see the non-joint branch)
          if constexpr (FINAL) gam += poison;
          size_t idx = ((size_t)b * K + k) * TS + tf + t;
          if (ok[0]) {
            if (a.out_aff) a.out_aff[idx] = gam;  // unmasked affiliation of this E-step
            if (a.out_q) a.out_q[idx] = q[0][k];
          }
          if constexpr (!FINAL) {
            double gs = ok[0] ? gam * sal : 0.0;
            // unconditional: a padding frame stores weight 0 (gs = 0, inv = 0), which is what the
            // unmasked M sweep needs -- and whatever parked data in wbuf since the last M phase
            L.wbuf[woff(k, tl[0])] = mweight(gs, q[0][k], invp);
            s[k] += gs;
          }
        }
      } else {
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const int t = tt[f];
        const double inv = L.inv_n2[tl[f]];
        // softmax over classes in mantissa/exponent form:
        //   exp(log_pdf_k) = 1 / (det_k q_k^D)   (cacg.py:200-201)
        // with q = m 2^e: one reciprocal of the mantissa serves both 1/q (M-step
        // weight) and q^-D = (1/m)^D 2^(-eD); 1/det_k is precomputed per class.
        double val[K], rq[K];
        int ex[K];
        int emax = INT32_MIN;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          double qq = fmax(fabs(q[f][k] * inv), kTiny);  // cacg.py:185-199
          q[f][k] = qq;
          int e;
          double m = frexp(qq, &e);
          double rm = fast_rcp(m);            // m in [0.5, 1)
          rq[k] = ldexp(rm, -e);              // 1/q (finite: q >= tiny)
          val[k] = rdet[k] * ipow<D>(rm);
          ex[k] = -(e * D + dete[k]);
          emax = max(emax, ex[k]);
        }
        double g[K], den = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          double w;
          if constexpr (TW) {
            w = w_kt ? __hip_atomic_load(w_kt + (size_t)k * TS + tf + t, __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_AGENT)
                     : a.in_weight[b * a.wb + k * a.wk + (int64_t)(tf + t) * a.wt];
          } else {
            w = wgt[k];
          }
          double v = ldexp(val[k], ex[k] - emax) * w;  // mixture_model_utils.py:32-37
          const uint8_t* act = FINAL ? a.final_activity : a.activity;
          if (act) v *= (double)act[((size_t)b * K + k) * TS + tf + t];
          g[k] = v;
          den += v;
        }
        // np.maximum and np.clip keep a NaN (mixture_model_utils.py:43-53), v_max / v_min drop it:
        // a non-finite class sum (NaN weights, a NaN model) would come out of the clip below as a
        // clean eps and the fit would report success where the reference's isfinite assert fires
        // (cacg.py:326-333).  `poison` is 0 for a finite sum and NaN otherwise; it rides on 1/|y|^2
        // into the M-step weights -> covariance -> PBBSS_ST_NONFINITE (two VALU per frame).
        const double poison = den - den;
        den = fmax(den, kTiny);  // mixture_model_utils.py:43-47
        const double rden = fast_rcp(den);
        const double sal = (!FINAL && a.saliency) ? a.saliency[(size_t)b * TS + tf + t] : 1.0;
        const double invp = inv + poison;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          double gam = g[k] * rden;
          // :50-53, no renormalisation.  In the loop the clip is applied unconditionally: with
          // eps = 0 it is [0, 1], which only catches a posterior that the reciprocal's rounding put
          // an ulp above 1 (the uniform `eps != 0` test cost two 64-bit selects per class)
          if (!FINAL || eps != 0.0) gam = fmin(fmax(gam, eps), 1.0 - eps);
          if constexpr (FINAL) {
            gam += poison;
            if (ok[f]) {
              size_t idx = ((size_t)b * K + k) * TS + tf + t;
              if (a.out_aff) a.out_aff[idx] = gam;
              if (a.out_q) a.out_q[idx] = q[f][k];
              if (a.out_logpdf) {
                // -D ln q - ln det B   (cacg.py:200-201); det = detm * 2^dete
                a.out_logpdf[idx] = -(double)D * log(q[f][k]) -
                                    (log(detm[k]) + (double)dete[k] * 0.6931471805599453);
              }
            }
          } else {
            double gs = ok[f] ? gam * sal : 0.0;
            if (pub_kt && ok[f])
              __hip_atomic_store(pub_kt + (size_t)k * TS + tf + t, gs, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
            // M-step weight gamma/max(q, 10 tiny)/|y|^2 (cacg.py:310, :322); q >= 10 tiny
            // except for all-zero frames, where inv = 0 makes the weight 0 anyway
            // (one v_min: 1/q <= 1/(10 tiny) exactly when q >= 10 tiny -- a compare and a 64-bit
            // select per class until round 5)
            double rqk = fmin(rq[k], 1.0 / (10.0 * kTiny));
            // unmasked for the first frame of a lane: a padding frame writes 0 (gs = 0, inv = 0,
            // rqk finite), which the unmasked M sweep relies on
            if (f == 0 || inl[f]) L.wbuf[woff(k, tl[f])] = gs * rqk * invp;
            s[k] += gs;
          }
        }
      }
      }  // !JOINT
    };
    {
      int t0 = 0;
      if constexpr (PAIR) {
        for (; a.T - t0 > kEmThreads; t0 += 2 * kEmThreads) pass(t0, std::integral_constant<int, 2>{});
      }
      for (; t0 < a.T; t0 += kEmThreads) pass(t0, std::integral_constant<int, 1>{});
    }
    if constexpr (!FINAL) wave_class_sums<K>(s, lane, L.red + wave * K);
  }

  // ---- phase M: wave W accumulates its share of the Hermitian entries ------
  // Entry slots of wave W within one class: its diagonal entries (rank s < NDW), then (re, im)
  // of its strict-upper pairs (rank s < NOW) -- kMap below says which.  All classes are
  // accumulated in one flat array acc[k * NSLOT + slot], padded to a multiple
  // of 16 for the halving butterfly.
  static constexpr int NSLOT = NDW + 2 * NOW;
  static constexpr int NACC = ((K * NSLOT + 15) / 16) * 16;

  // Which wave accumulates which entry.  Default: round robin (diagonal i -> wave i % 4, pair p ->
  // wave p % 4).  D = 8 (round 5): the 28 pairs are grouped by CHANNELS -- waves 0 / 1 take the
  // pairs inside {0,1,2,3} / {4,5,6,7} plus one cross pair, waves 2 / 3 the remaining cross pairs
  // of {0,1} / {2,3} with {4,..,7} -- so that a wave needs 5 or 6 of the 8 channels: three of the
  // four float4 planes of a frame to load and 10 or 12 instead of 16 values to widen per trip
  // (round robin touches every channel in every wave).  7 pairs + 2 diagonals per wave as before.
  struct EntryMap {
    int dwave[D], drank[D];                        // diagonal i -> wave, rank among the wave's
    int pwave[NOFF > 0 ? NOFF : 1], prank[NOFF > 0 ? NOFF : 1];  // pair p -> wave, rank
    unsigned long long dcode[kEmWaves], pcode[kEmWaves];  // inverse: 4 / 5 bits per rank, all ones = none
  };
  static constexpr EntryMap make_entry_map() {
    EntryMap m{};
    int nd[kEmWaves] = {}, np[kEmWaves] = {};
    for (int w = 0; w < kEmWaves; ++w) {
      m.dcode[w] = ~0ull;
      m.pcode[w] = ~0ull;
    }
    for (int i = 0; i < D; ++i) {
      int w = i % kEmWaves;
      if (D == 8) w = (i < 2) ? 0 : (i < 4) ? 3 : (i < 6) ? 1 : 2;
      m.dwave[i] = w;
      m.drank[i] = nd[w]++;
    }
    for (int p = 0; p < NOFF; ++p) {
      const int i = tri_i<D>(p), j = tri_j<D>(p);
      int w = p % kEmWaves;
      if (D == 8) {
        if (j < 4) w = 0;                                  // inside {0,1,2,3}
        else if (i >= 4) w = 1;                            // inside {4,5,6,7}
        else if (i == 0 && j == 4) w = 0;                  // the two cross pairs that fill waves 0 / 1
        else if (i == 3 && j == 7) w = 1;
        else w = (i < 2) ? 2 : 3;                          // {0,1} x {4..7} / {2,3} x {4..7}
      }
      m.pwave[p] = w;
      m.prank[p] = np[w]++;
    }
    for (int i = 0; i < D; ++i) {
      const int w = m.dwave[i], r = m.drank[i];
      m.dcode[w] = (m.dcode[w] & ~(15ull << (4 * r))) | ((unsigned long long)i << (4 * r));
    }
    for (int p = 0; p < NOFF; ++p) {
      const int w = m.pwave[p], r = m.prank[p];
      m.pcode[w] = (m.pcode[w] & ~(31ull << (5 * r))) | ((unsigned long long)p << (5 * r));
    }
    return m;
  }
  static constexpr EntryMap kMap = make_entry_map();
  static constexpr bool entry_map_fits() {
    for (int i = 0; i < D; ++i)
      if (kMap.drank[i] >= NDW) return false;
    for (int p = 0; p < NOFF; ++p)
      if (kMap.prank[p] >= NOW) return false;
    return true;
  }
  static_assert(entry_map_fits(), "a wave owns more entries than it has accumulator slots");

  // ---- write-back table of the M phase ----------------------------------------------------
  // After the halving butterfly lane group g = (lane >> 2) & 15 of wave W holds the totals of the
  // accumulator indices R g .. R g + R - 1 (R = NACC / 16).  Where each of them goes in cpack
  // (class, packed entry) depends on the lane only: computed ONCE per kernel into LDS by carve()
  // (until round 5 every wave decoded it after every butterfly: ~70 branchy instructions per
  // wave-iteration).  Values that belong to no entry go to `sink`.
  static constexpr int kWbR = NACC / 16;
  static __host__ __device__ constexpr size_t wbtab_bytes() {
    return ((size_t)kEmWaves * 16 * kWbR * sizeof(unsigned short) + 7) & ~(size_t)7;
  }
  // ROUND_ROBIN: the entry map of the packed-FP32 kernel (diagonal i -> wave i % 4, pair p -> p % 4)
  template <bool ROUND_ROBIN = false>
  static __device__ __forceinline__ void fill_wbtab(const Lds& L) {
    const int tid = threadIdx.x;
    if (tid >= kEmWaves * 16) return;
    const int w = tid >> 4, g = tid & 15;
    const unsigned long long dcode = kMap.dcode[w], pcode = kMap.pcode[w];
    const int sink = (int)(L.sink - L.cpack);
#pragma unroll
    for (int m = 0; m < kWbR; ++m) {
      const int idx = kWbR * g + m;
      int off = sink;
      if (idx < K * NSLOT) {
        const int k = idx / NSLOT, sl = idx % NSLOT;
        const bool dg = sl < NDW;
        const int u = ROUND_ROBIN
                          ? (dg ? sl * kEmWaves + w : ((sl - NDW) >> 1) * kEmWaves + w)
                          : (dg ? (int)((dcode >> (4 * sl)) & 15)
                                : (int)((pcode >> (5 * ((sl - NDW) >> 1))) & 31));
        const int e = dg ? u : D + 2 * u + ((sl - NDW) & 1);
        if (u < (dg ? D : NOFF)) off = k * NA + e;
      }
      L.wbtab[(w * 16 + g) * kWbR + m] = (unsigned short)off;
    }
  }

  // c_begin / c_end: the chunks [c_begin, c_end) of the frame arrays only (c_end < 0: all of
  // them) and cpack_out: where the packed sums go (null: L.cpack) -- the eight-wave Watson kernel
  // splits the frames of a bin over two groups of four waves and adds the two partial arrays
  template <int W>
  static __device__ void phase_m(const EmArgs& a, const Lds& L, int lane, int c_begin = 0,
                                 int c_end = -1, double* cpack_out = nullptr) {
    lane = opaque(lane);
    double acc[NACC];
    // One trip = one chunk of 64 frames, lane = frame, whole chunks only: the padding frames of
    // the last chunk carry w = 0 and y = 0.  The lane's addresses advance by one chunk per trip;
    // the planes of a chunk (channel pairs, classes) sit at compile-time offsets.  The first
    // trip initialises the accumulators with products instead of adding to zeros.
    const YS* yp = L.ybuf + (size_t)lane * 4 + (size_t)c_begin * (DP * kFC * 4);
    const double* wp = L.wbuf + lane + (size_t)c_begin * (K * kFC);
    // raw frame + weights of one chunk: loaded one trip ahead of their use (PBBSS_M_PREFETCH), so
    // that the LDS round trip of chunk c + 1 hides behind the 96 FMAs of chunk c
    struct Raw {
      YS4 v[DP];
      double w[K];
    };
    auto fetch = [&](Raw& r) {
      static_for<0, DP>([&](auto dpc) {
        constexpr int dp = dpc;
        r.v[dp] = *reinterpret_cast<const YS4*>(yp + dp * (kFC * 4));
      });
#pragma unroll
      for (int k = 0; k < K; ++k) r.w[k] = wp[k * kFC];
      yp += DP * kFC * 4;
      wp += K * kFC;
    };
    auto compute = [&](const Raw& r, auto firstc) {
      constexpr bool FIRST = firstc;
      double re[D], im[D];
      static_for<0, DP>([&](auto dpc) {
        constexpr int dp = dpc;
        re[2 * dp] = (double)r.v[dp].x;
        im[2 * dp] = (double)r.v[dp].y;
        if constexpr (2 * dp + 1 < D) {
          re[2 * dp + 1] = (double)r.v[dp].z;
          im[2 * dp + 1] = (double)r.v[dp].w;
        }
      });
      const double (&w)[K] = r.w;
      static_for<0, D>([&](auto ic) {
        constexpr int i = ic;
        if constexpr (kMap.dwave[i] == W) {
          double dg = re[i] * re[i] + im[i] * im[i];
#pragma unroll
          for (int k = 0; k < K; ++k) {
            double& x = acc[k * NSLOT + kMap.drank[i]];
            x = FIRST ? w[k] * dg : fma(w[k], dg, x);
          }
        }
      });
      static_for<0, NOFF>([&](auto pc) {
        constexpr int p = pc;
        if constexpr (kMap.pwave[p] == W) {
          constexpr int i = tri_i<D>(p), j = tri_j<D>(p);
          constexpr int s = NDW + 2 * kMap.prank[p];
          double pr = re[i] * re[j] + im[i] * im[j];
          double pim = im[i] * re[j] - re[i] * im[j];
#pragma unroll
          for (int k = 0; k < K; ++k) {
            double& xr = acc[k * NSLOT + s];
            double& xi = acc[k * NSLOT + s + 1];
            xr = FIRST ? w[k] * pr : fma(w[k], pr, xr);
            xi = FIRST ? w[k] * pim : fma(w[k], pim, xi);
          }
        }
      });
    };
    // slots no entry maps to (wave W owns fewer diagonal entries / pairs than NDW / NOW; padding
    // of the butterfly) are never touched by a trip
    static_for<0, NACC>([&](auto xc) {
      constexpr int x = xc;
      constexpr int sl = x % NSLOT;
      constexpr bool used =
          x < K * NSLOT &&
          (sl < NDW ? (int)((kMap.dcode[W] >> (4 * sl)) & 15) < D
                    : (int)((kMap.pcode[W] >> (5 * ((sl - NDW) >> 1))) & 31) < NOFF);
      if constexpr (!used) acc[x] = 0.0;
    });
    const int nchunk = (c_end < 0 ? (padded_frames(a.T) >> 6) : c_end) - c_begin;  // >= 1
#if PBBSS_M_PREFETCH
    {
      // two buffers, ping-pong (no copies): the loads of chunk c + 1 are in flight while chunk c
      // is accumulated; LDS returns in order, so the wait in front of a compute only covers its
      // own buffer
      Raw A, B;
      fetch(A);
      if (nchunk > 1) {
        fetch(B);
        __builtin_amdgcn_sched_barrier(0);
        compute(A, std::true_type{});
        int c = nchunk - 1;  // chunks left to accumulate, B (loaded) included
        while (c > 2) {
          fetch(A);
          __builtin_amdgcn_sched_barrier(0);
          compute(B, std::false_type{});
          fetch(B);
          __builtin_amdgcn_sched_barrier(0);
          compute(A, std::false_type{});
          c -= 2;
        }
        if (c == 2) {
          fetch(A);
          __builtin_amdgcn_sched_barrier(0);
          compute(B, std::false_type{});
          compute(A, std::false_type{});
        } else {
          compute(B, std::false_type{});
        }
      } else {
        compute(A, std::true_type{});
      }
    }
#else
    {
      Raw r;
      fetch(r);
      compute(r, std::true_type{});
      for (int c = nchunk; c > 1; --c) {
        fetch(r);
        compute(r, std::false_type{});
      }
    }
#endif
#ifdef PBBSS_PHASE_PROFILE
    unsigned long long tm0 = __builtin_readcyclecounter(), tm1 = tm0;
#endif
    // halving butterfly over the 64 frame-lanes; 16 lanes each store NACC/16 totals
    // Issue priority (round 6).  The butterfly and the factorisation are chains of dependent
    // instructions; the trips above and the E phase are issue-bound streams.  When a chain shares
    // its SIMD with another workgroup's stream at equal priority it waits for issue slots it
    // could use at once and then leaves the SIMD to the stream while its result is in flight:
    // s_setprio 1 for the butterfly + write-back, 3 for the factorisation, 0 again afterwards --
    // 1.304 -> 1.257 ms per fit on one box (profiles/r06_e_issue_priority.txt).  Member
    // workgroups of a split group keep the level of their launch (split_prio) throughout.
    if (a.split_groups == 0) __builtin_amdgcn_s_setprio(1);
    wave_reduce_scatter<NACC>(acc, lane);
#ifdef PBBSS_PHASE_PROFILE
    tm1 = __builtin_readcyclecounter();
#endif
    // write-back in the packed order of apack (diag i -> i, pair p -> D + 2p + {Re, Im}) through
    // the per-lane table carve() left in LDS; the consumers (factor_class, split_exchange, psd
    // read-out) unpack with pair_index()
    if ((lane & 3) == 0) {
      const unsigned short* tab = L.wbtab + (W * 16 + ((lane >> 2) & 15)) * kWbR;
      double* dst = cpack_out ? cpack_out : L.cpack;
#pragma unroll
      for (int m = 0; m < kWbR; ++m) dst[tab[m]] = acc[m];
    }
    if (a.split_groups == 0) __builtin_amdgcn_s_setprio(0);
#ifdef PBBSS_PHASE_PROFILE
    if (a.prof && lane == 0 && W == 0) {
      unsigned long long tm2 = __builtin_readcyclecounter();
      atomicAdd(a.prof + 64, tm1 - tm0);  // butterfly
      atomicAdd(a.prof + 65, tm2 - tm1);  // write-back
      atomicAdd(a.prof + 66, 1ull);
    }
#endif
  }

  // packed index of pair (i < j)
  static __device__ __forceinline__ int pair_index(int i, int j) {
    return i * D - (i * (i + 1)) / 2 + (j - i - 1);
  }

  // covariance sum C_k[i][j] from the packed LDS array (C_ji = conj(C_ij))
  static __device__ __forceinline__ void cov_entry(const Lds& L, int k, int i, int j, double& re,
                                                   double& im) {
    if (i == j) {
      re = L.cpack[k * NA + i];
      im = 0.0;
    } else {
      const int lo = min(i, j), hi = max(i, j);
      const double2 v = *reinterpret_cast<const double2*>(L.cpack + k * NA + D + 2 * pair_index(lo, hi));
      re = v.x;
      im = (i < j) ? v.y : -v.y;
    }
  }

  // store Hermitian G (lane = entry) as the dot-ready A_k
  static __device__ __forceinline__ void store_apack(const Lds& L, int k, LaneIJ c, double gre,
                                                     double gim) {
    if (c.i < D && c.j < D) {
      if (c.i == c.j) {
        L.apack[k * NA + c.i] = gre;
      } else if (c.i < c.j) {
        int p = pair_index(c.i, c.j);
        L.apack[k * NA + D + 2 * p] = 2.0 * gre;
        L.apack[k * NA + D + 2 * p + 1] = 2.0 * gim;
      }
    }
  }

  // A = V diag(1/lam) V^H, V_ij on lanes, lam_col = eigenvalue of this lane's column
  static __device__ __forceinline__ void inverse_from_eig(double vre, double vim, double lam_col,
                                                          LaneIJ c, double& gre, double& gim) {
    gre = 0.0;
    gim = 0.0;
#pragma unroll
    for (int e = 0; e < D; ++e) {
      double il = 1.0 / lane_get(lam_col, ij_lane(0, e));
      double ar = lane_get(vre, ij_lane(c.i, e)), ai = lane_get(vim, ij_lane(c.i, e));
      double br = lane_get(vre, ij_lane(c.j, e)), bi = lane_get(vim, ij_lane(c.j, e));
      // V_ie conj(V_je) / lam_e
      gre += (ar * br + ai * bi) * il;
      gim += (ai * br - ar * bi) * il;
    }
  }

  // ---- phase F for one class (one wave) ------------------------------------
  static __device__ void factor_class(const EmArgs& a, const Lds& L, int64_t b, int k, int lane,
                                      bool last) {
    // a dependent chain beside another workgroup's issue-bound stream: top issue priority for its
    // duration (see phase_m; members of a split group keep the level of their launch)
    if (a.split_groups == 0) __builtin_amdgcn_s_setprio(3);
    lane = opaque(lane);
    const LaneIJ c = lane_ij(lane);
    const bool valid = c.i < D && c.j < D;
#ifdef PBBSS_PHASE_PROFILE
    unsigned long long fs[6] = {0, 0, 0, 0, 0, 0};
    int fsn = 0;
#define PBBSS_FSTAMP fs[fsn++] = __builtin_readcyclecounter();
    PBBSS_FSTAMP
#else
#define PBBSS_FSTAMP
#endif
    // covariance entry of this lane first: the LDS round trip overlaps the class sums below
    double are = 0.0, aim = 0.0;
    if (valid) cov_entry(L, k, c.i, c.j, are, aim);
    // class sums of the E phase (per-wave partials in L.red) -> sum_t gamma_kt and the
    // new mixture weight (mixture_model_utils.py:133-203); done here by the wave that
    // owns class k instead of a separate single-thread step between two barriers
    double S = 0.0, tot = 0.0;
#pragma unroll
    for (int kk = 0; kk < K; ++kk) {
      double sk = 0.0;
#pragma unroll
      for (int w = 0; w < kEmWaves; ++w) sk += L.red[w * K + kk];
      tot += fabs(sk);
      S = (kk == k) ? sk : S;
    }
    // D / max(S, tiny) (cacg.py:316, :327) by a Newton-refined reciprocal (~1 ulp) instead of an
    // IEEE division sequence: this sits on the serial path of every iteration
    const double scale = (double)D * fast_rcp(fmax(S, kTiny));
    are *= scale;
    aim *= scale;
    int st = 0;
    // Non-finite input (cacg.py:333) is detected on the slow path only: a NaN / Inf anywhere in a
    // Hermitian matrix reaches a later Gauss-Jordan pivot (a_jj -= a_ji a_ij / d), which fails the
    // pivot test below and sends the class to the exact eigen path, where the flag is raised.
    if (last && a.out_cov && valid) {
      double* oc = a.out_cov + ((((size_t)b * K + k) * D + c.i) * D + c.j) * 2;
      oc[0] = are;
      oc[1] = aim;
    }
    bool need_eig = last || a.force_eig || (*L.flags & 1);
    PBBSS_FSTAMP
    if (!need_eig) {
      double gre = are, gim = aim;
      ScaledReal det;
      // trace of C: independent of the inversion, so its reduction fills the sweep's stalls
      double trc = wave_sum((valid && c.i == c.j) ? are : 0.0);
      int info = wave_hpd_inverse<D>(gre, gim, c, det);
      bool ok = (info == 0);
      PBBSS_FSTAMP
      if (ok) {
        // the result is stored right away (the slow path below overwrites it in the rare case
        // the bound fails), so the stores overlap the reduction of the bound
        store_apack(L, k, c, gre, gim);
        if (lane == 0) {
          L.detm[k] = det.m;
          L.rdet[k] = fast_rcp(det.m);  // mantissa in [0.5, 1)
          L.dete[k] = det.e;
        }
        // lambda_min >= 1/||A^-1||_F and lambda_max <= tr C: if even this pessimistic
        // ratio stays clear of the floor, no eigenvalue is floored (cacg.py:112-126).
        // Compared as squares (no square root): bound^2 = tr^2 * ||A^-1||_F^2.
        double fro2 = wave_sum(valid ? gre * gre + gim * gim : 0.0);
        const double bound2 = trc * trc * fro2;
        ok = isfinite(bound2) && (bound2 * (a.eig_floor * a.eig_floor) < 1e-4) && (bound2 < 1e26);
      }
      if (!ok) {
        need_eig = true;
        st |= PBBSS_ST_SLOWPATH;
      }
      PBBSS_FSTAMP
#ifdef PBBSS_PHASE_PROFILE
      if (a.prof && lane == 0 && k == 0 && !need_eig) {
        atomicAdd(a.prof + 67, fs[1] - fs[0]);  // sums, scale, load
        atomicAdd(a.prof + 68, fs[2] - fs[1]);  // trace + Gauss-Jordan
        atomicAdd(a.prof + 69, fs[3] - fs[2]);  // bound, store
        atomicAdd(a.prof + 70, 1ull);
      }
#endif
    }
    if (need_eig) {
      const bool finite_in = isfinite(are) && isfinite(aim);
      if (wave_or(finite_in ? 0 : 1)) st |= PBBSS_ST_NONFINITE;  // cacg.py:333
    }
    if (need_eig) {
      if (a.covariance_norm == PBBSS_COVNORM_TRACE) {  // cacg.py:88-90
        double tr = wave_sum((valid && c.i == c.j) ? are : 0.0);
        double it = 1.0 / fmax(tr, kTiny);
        are *= it;
        aim *= it;
      }
      double vre, vim;
      int sweeps = wave_jacobi_heev<D>(are, aim, c, vre, vim);
      if (sweeps < 0) st |= PBBSS_ST_EIG_NOCONV;
      // eigenvalue of this lane's column, broadcast from the diagonal
      double lam = lane_get(are, ij_lane(c.j, c.j));
      double lmax = wave_max((c.i == 0 && c.j < D) ? lam : -1.79e308);
      double lout;
      if (a.covariance_norm == PBBSS_COVNORM_EIGENVALUE) {  // cacg.py:112-121
        lout = lam / fmax(lmax, kTiny);
        if (lout < a.eig_floor) {
          lout = a.eig_floor;
          if (c.i == 0 && c.j < D) st |= PBBSS_ST_FLOORED;
        }
      } else {  // cacg.py:122-126
        double fl = lmax * a.eig_floor;
        lout = lam;
        if (lout < fl) {
          lout = fl;
          if (c.i == 0 && c.j < D) st |= PBBSS_ST_FLOORED;
        }
      }
      if (c.i == 0 && c.j < D && !isfinite(lout)) st |= PBBSS_ST_NONFINITE;  // cacg.py:127
      st = wave_or(st);
      if (c.j >= D) lout = 1.0;  // padding columns: harmless in the products below
      if (last) {
        // numpy.linalg.eigh order: ascending eigenvalues, eigenvectors in columns
        int rank = wave_sort_rank<D>(lam, c);
        if (valid) {
          if (a.out_eigvec) {
            double* ov = a.out_eigvec + ((((size_t)b * K + k) * D + c.i) * D + rank) * 2;
            ov[0] = vre;
            ov[1] = vim;
          }
          if (a.out_eigval && c.i == 0) a.out_eigval[((size_t)b * K + k) * D + rank] = lout;
        }
      }
      double gre, gim;
      inverse_from_eig(vre, vim, lout, c, gre, gim);
      store_apack(L, k, c, gre, gim);
      ScaledReal det{1.0, 0};
#pragma unroll
      for (int e = 0; e < D; ++e) scaled_mul(det, fmax(lane_get(lout, ij_lane(0, e)), kTiny));
      if (lane == 0) {
        L.detm[k] = det.m;
        L.rdet[k] = 1.0 / det.m;
        L.dete[k] = det.e;
      }
    }
    if (lane == 0) {
      L.status[k] |= st;
      // new mixture weight (off the serial path of the factorisation)
      double wnew;
      if (a.weight_mode >= PBBSS_WEIGHT_SHARED_K) {
        wnew = L.wgt[k];  // weights shared across problems: run_shared sets them
      } else if (a.weight_mode == PBBSS_WEIGHT_UNIFORM) {
        wnew = 1.0 / K;  // mixture_model_utils.py:180-183
      } else if (a.saliency) {
        wnew = S / ((tot == 0.0) ? 1e-10 : tot);  // :192-201
      } else {
        wnew = S / (double)t_stride(a);  // :188
      }
      L.wgt[k] = wnew;
    }
    if (a.split_groups == 0) __builtin_amdgcn_s_setprio(0);
  }

  // model (V, lambda) given by the caller -> A_k, det  (cacgmm.py:229-234)
  static __device__ void prep_from_model(const EmArgs& a, const Lds& L, int64_t b, int k,
                                         int lane) {
    const LaneIJ c = lane_ij(lane);
    const bool valid = c.i < D && c.j < D;
    double vre = 0.0, vim = 0.0, lam = 1.0;
    if (valid) {
      const double* v = a.in_eigvec + ((((size_t)b * K + k) * D + c.i) * D + c.j) * 2;
      vre = v[0];
      vim = v[1];
      lam = a.in_eigval[((size_t)b * K + k) * D + c.j];
    }
    double gre, gim;
    inverse_from_eig(vre, vim, lam, c, gre, gim);
    store_apack(L, k, c, gre, gim);
    ScaledReal det{1.0, 0};
#pragma unroll
    for (int e = 0; e < D; ++e) scaled_mul(det, fmax(lane_get(lam, ij_lane(0, e)), kTiny));
    if (lane == 0) {
      L.detm[k] = det.m;
      L.rdet[k] = 1.0 / det.m;
      L.dete[k] = det.e;
      // loop E-steps read wgt from LDS; a (b,k)-strided weight is enough there
      L.wgt[k] = a.in_weight ? a.in_weight[b * a.wb + k * a.wk] : 1.0 / K;
    }
  }

  // ===================== joint spatial + spectral models ======================
  // q_k(t) = <A_k, P_t> / |y_t|^2 without the operand pipelining of phase_e (used
  // once per launch by the inline permutation search below)
  static __device__ __forceinline__ void quad_forms(const Lds& L, int t, double (&q)[K]) {
    double re[D], im[D];
    load_frame(L, t, re, im);
#pragma unroll
    for (int k = 0; k < K; ++k) q[k] = 0.0;
    static_for<0, D>([&](auto ic) {
      constexpr int i = ic;
      double dg = re[i] * re[i] + im[i] * im[i];
#pragma unroll
      for (int k = 0; k < K; ++k) q[k] = fma(L.apack[k * NA + i], dg, q[k]);
    });
    static_for<0, NOFF>([&](auto pc) {
      constexpr int p = pc;
      constexpr int i = tri_i<D>(p), j = tri_j<D>(p);
      double pr = re[i] * re[j] + im[i] * im[j];
      double pim = im[i] * re[j] - re[i] * im[j];
#pragma unroll
      for (int k = 0; k < K; ++k)
        q[k] = fma(L.apack[k * NA + D + 2 * p], pr, fma(L.apack[k * NA + D + 2 * p + 1], pim, q[k]));
    });
    const double inv = L.inv_n2[t];
#pragma unroll
    for (int k = 0; k < K; ++k) q[k] = fmax(fabs(q[k] * inv), kTiny);
  }

  // <A_k, P_t> for one frame with the operand pipelining of phase_e (A_k of chunk c+1 fetched
  // from LDS before the FMAs of chunk c issue; compiler fences pin that order -- left alone
  // hipcc front-loads all K*D*D broadcast reads of a pass and spills them).  Unscaled.
  static __device__ __forceinline__ void quad_forms_pipelined(const Lds& L, const double (&re)[D],
                                                              const double (&im)[D],
                                                              double (&q)[K]) {
    constexpr int NCH = (NOFF + kOperandChunk - 1) / kOperandChunk;
    double op[2][kOperandChunk][K][2];
    auto fetch = [&](auto cc, auto bb) {
      constexpr int c = cc, bsel = bb;
#pragma unroll
      for (int x = 0; x < kOperandChunk; ++x) {
        if (c * kOperandChunk + x < NOFF) {
#pragma unroll
          for (int k = 0; k < K; ++k) {
            op[bsel][x][k][0] = L.apack[k * NA + D + 2 * (c * kOperandChunk + x)];
            op[bsel][x][k][1] = L.apack[k * NA + D + 2 * (c * kOperandChunk + x) + 1];
          }
        }
      }
    };
#pragma unroll
    for (int k = 0; k < K; ++k) q[k] = 0.0;
    if constexpr (NCH > 0) fetch(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    asm volatile("" ::: "memory");
    static_for<0, D>([&](auto ic) {
      constexpr int i = ic;
      double ad[K];
#pragma unroll
      for (int k = 0; k < K; ++k) ad[k] = L.apack[k * NA + i];
      double dg = re[i] * re[i] + im[i] * im[i];
#pragma unroll
      for (int k = 0; k < K; ++k) q[k] = fma(ad[k], dg, q[k]);
    });
    static_for<0, NCH>([&](auto cc) {
      constexpr int c = cc;
      constexpr int cur = c & 1;
      if constexpr (c + 1 < NCH) {
        fetch(std::integral_constant<int, c + 1>{}, std::integral_constant<int, 1 - cur>{});
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, kOperandChunk>([&](auto xc) {
        constexpr int x = xc;
        constexpr int p = c * kOperandChunk + x;
        if constexpr (p < NOFF) {
          constexpr int i = tri_i<D>(p), j = tri_j<D>(p);
          double pr = re[i] * re[j] + im[i] * im[j];
          double pim = im[i] * re[j] - re[i] * im[j];
#pragma unroll
          for (int k = 0; k < K; ++k)
            q[k] = fma(op[cur][x][k][0], pr, fma(op[cur][x][k][1], pim, q[k]));
        }
      });
      // tie the partial sums to the fence: the FMAs are pure arithmetic, which instruction
      // selection may otherwise sink below all later fences (then every prefetched operand
      // is spilled until the arithmetic finally runs)
#pragma unroll
      for (int k = 0; k < K; ++k) asm volatile("" : "+v"(q[k])::"memory");
      __builtin_amdgcn_sched_barrier(0);
    });
  }

  // p-th permutation of (0..K-1) in lexicographic order (itertools.permutations order)
  static __device__ __forceinline__ void nth_permutation(int p, int (&perm)[K]) {
    int fact = 1;
#pragma unroll
    for (int i = 2; i < K; ++i) fact *= i;  // (K-1)!
    unsigned used = 0;
#pragma unroll
    for (int i = 0; i < K; ++i) {
      int d = p / fact;
      p -= d * fact;
      if (i < K - 1) fact /= (K - 1 - i);
      int pick = 0;
#pragma unroll
      for (int c = 0, seen = 0; c < K; ++c) {
        if (!(used >> c & 1)) {
          if (seen == d) pick = c;
          ++seen;
        }
      }
      used |= 1u << pick;
      perm[i] = pick;
    }
  }

  // Inline permutation alignment of integration models (mixture_model_utils.py:58-130):
  // per frequency bin, the class permutation of the spatial log-pdf that maximises
  // sum_{k,t} softmax_k(lp)(t) * lp_k(t), lp = spatial[perm] + spectral (no weights).
  // The weighted spatial log-pdf is parked in wbuf (free until the E-step).
  static __device__ void phase_joint_pa(const EmArgs& a, const Lds& L, int64_t b, int tid,
                                        int wave, int lane, const JointExtras& jx, int* perm_out) {
    for (int t = tid; t < a.T; t += kEmThreads) {
      double q[K];
      quad_forms(L, t, q);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        double ld = log(L.detm[k]) + (double)L.dete[k] * 0.6931471805599453;
        L.wbuf[woff(k, t)] = jx.spatial_weight * (-(double)D * log(q[k]) - ld);
      }
    }
    __syncthreads();
    int nperm = 1;
#pragma unroll
    for (int i = 2; i <= K; ++i) nperm *= i;
    double best = -INFINITY;
    int best_p = 0;
    for (int p = 0; p < nperm; ++p) {
      int perm[K];
      nth_permutation(p, perm);
      double part = 0.0;
      for (int t = tid; t < a.T; t += kEmThreads) {
        double lp[K], mx = -1.79e308;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          lp[k] = L.wbuf[woff(perm[k], t)] +
                  jx.extra_logpdf[((size_t)b * K + k) * a.T + t];
          mx = fmax(mx, lp[k]);
        }
        double e[K], den = 0.0, num = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          e[k] = exp(lp[k] - mx);
          den += e[k];
          num = fma(e[k], lp[k], num);
        }
        part += num / fmax(den, kTiny);
      }
      part = wave_sum(part);
      __syncthreads();  // previous round's reads of red are done
      if (lane == 0) L.red[wave] = part;
      __syncthreads();
      double tot = 0.0;
#pragma unroll
      for (int w = 0; w < kEmWaves; ++w) tot += L.red[w];
      if (tot > best) {  // strict: the first maximiser wins, as in the reference loop
        best = tot;
        best_p = p;
      }
    }
    __syncthreads();
    if (tid == 0) {
      int perm[K];
      nth_permutation(best_p, perm);
#pragma unroll
      for (int k = 0; k < K; ++k) perm_out[k] = perm[k];
    }
    __syncthreads();
  }

  // One joint EM iteration for the cACG half (a.iterations == 1): E-step with the old
  // model + spectral log-pdf -> affiliation (and q) to HBM, covariance update, eigen
  // decomposition.  a.iterations == 0: the E-step only (model.predict).
  // Mixture weights are always read through (wb, wk, wt) from a.in_weight.
  template <bool PA>
  static __device__ void run_joint(const EmArgs& a, const JointExtras& jx0, char* smem) {
    static_assert(!SPILL, "joint kernels keep the observation in LDS");
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    if constexpr (!PA) {  // members are never launched together with the inline aligner
      if (jx0.main_grid > 0 && (int)blockIdx.x >= jx0.main_grid) {
        run_joint_member(a, jx0, smem);
        return;
      }
    }
    const int bstride = jx0.main_grid > 0 ? jx0.main_grid : (int)gridDim.x;
    const Lds L = carve(smem, a.T);
    int* perm = reinterpret_cast<int*>(smem + lds_bytes(a.T));
    for (int64_t b = blockIdx.x; b < a.B; b += bstride) {
      __syncthreads();
      if (tid < K) L.status[tid] = 0;
      if (tid == 0) *L.flags = 0;
      __syncthreads();
      phase_load(a, L, b, tid);
      __syncthreads();
      for (int k = wave; k < K; k += kEmWaves) {
        if (jx0.state_in) {
          const double* st = jx0.state_in + ((size_t)b * K + k) * (NA + 2);
          for (int i = lane; i < NA; i += kWave) L.apack[k * NA + i] = st[i];
          if (lane == 0) {
            const double dm = st[NA];
            L.detm[k] = dm;
            L.rdet[k] = 1.0 / dm;
            L.dete[k] = (int)st[NA + 1];
          }
        } else {
          prep_from_model(a, L, b, k, lane);
        }
      }
      __syncthreads();
      JointExtras jx = jx0;
      jx.perm = nullptr;
      if constexpr (PA) {
        phase_joint_pa(a, L, b, tid, wave, lane, jx, perm);
        jx.perm = perm;
      }
      if (a.iterations == 0) {
        phase_e<true, true, true, false, PA>(a, L, b, tid, wave, lane, a.final_eps, 0, &jx);
        continue;
      }
      phase_e<false, true, true, false, PA>(a, L, b, tid, wave, lane, a.aff_eps, 0, &jx);
      __syncthreads();
      if (jx0.weight_fk_out && tid == 0) {
        double v[K], tot = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          v[k] = 0.0;
#pragma unroll
          for (int w = 0; w < kEmWaves; ++w) v[k] += L.red[w * K + k];
          tot += v[k];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) jx0.weight_fk_out[(size_t)b * K + k] = v[k] / tot;
      }
      switch (wave) {
        case 0: phase_m<0>(a, L, lane); break;
        case 1: phase_m<1>(a, L, lane); break;
        case 2: phase_m<2>(a, L, lane); break;
        default: phase_m<3>(a, L, lane); break;
      }
      __syncthreads();
      for (int k = wave; k < K; k += kEmWaves) {
        factor_class(a, L, b, k, lane, jx0.emit_model != 0);
        if (jx0.state_out) {  // same wave wrote apack / det of class k: wave-ordered LDS
          double* st = jx0.state_out + ((size_t)b * K + k) * (NA + 2);
          for (int i = lane; i < NA; i += kWave) st[i] = L.apack[k * NA + i];
          if (lane == 0) {
            st[NA] = L.detm[k];
            st[NA + 1] = (double)L.dete[k];
          }
        }
      }
      __syncthreads();
      if (tid < K && a.out_status) a.out_status[(size_t)b * K + tid] = L.status[tid];
    }
  }

  // ---- phase_load and phase_init_gamma in one pass (spatial kernel of the rotated joint loop):
  // layout TD, gamma0 / q0 given.  Same arithmetic as the two phases, statement by statement.
  static __device__ void phase_load_gamma(const EmArgs& a, const Lds& L, int64_t b, int tid,
                                          int wave, int lane) {
    const int T = a.T;
    const YS2* yg = reinterpret_cast<const YS2*>(a.y);
    bool zero_seen = false;
    double s[K];
#pragma unroll
    for (int k = 0; k < K; ++k) s[k] = 0.0;
    for (int t = tid; t < L.Tp; t += kEmThreads) {
      const int tc = t < T ? t : T - 1;  // clamped: unconditional loads, masked below
      YS2 v[D];
      double g[K], q[K];
#pragma unroll
      for (int d = 0; d < D; ++d) v[d] = yg[((size_t)b * T + tc) * D + d];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const size_t idx = ((size_t)b * K + k) * T + tc;
        g[k] = a.gamma0[idx];
        q[k] = a.q0[idx];
      }
      const double sal = a.saliency ? a.saliency[(size_t)b * T + tc] : 1.0;
      double n2 = 0.0;
      YS vr[2 * DP], vi[2 * DP];
#pragma unroll
      for (int d = 0; d < 2 * DP; ++d) {
        vr[d] = (YS)0;
        vi[d] = (YS)0;
      }
      if (t < T) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
          vr[d] = v[d].x;
          vi[d] = v[d].y;
          n2 += (double)v[d].x * (double)v[d].x + (double)v[d].y * (double)v[d].y;
        }
      }
#pragma unroll
      for (int dp = 0; dp < DP; ++dp) {
        YS4 o;
        o.x = vr[2 * dp];
        o.y = vi[2 * dp];
        o.z = vr[2 * dp + 1];
        o.w = vi[2 * dp + 1];
        *reinterpret_cast<YS4*>(L.ybuf + yoff(dp, t)) = o;
      }
      const double inv = (t < T) ? ((n2 > 0.0) ? 1.0 / n2 : 0.0) : 0.0;
      L.inv_n2[t] = inv;
      if (t < T && !(n2 > 0.0)) zero_seen = true;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const double gs = (t < T) ? g[k] * sal : 0.0;
        L.wbuf[woff(k, t)] = (t < T) ? mweight(gs, q[k], inv) : 0.0;  // padding frames: weight 0
        s[k] += gs;
      }
    }
    if (zero_seen) atomicOr(L.flags, 1);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      double tot = wave_sum(s[k]);
      if (lane == 0) L.red[wave * K + k] = tot;
    }
  }

  // ---- quadratic forms of the model in LDS for every frame -> q_out (rotated joint loop) ----
  // The dot product <A_k, P_t> of phase_e (DPP operands out of one register per 16 entries),
  // one frame per lane; the clamp is the reference's (cacg.py:185-199).
  static __device__ void phase_q(const EmArgs& a, const Lds& L, int64_t b, int tid, int lane,
                                 double* q_out, int tf = 0) {
    tid = opaque(tid);
    lane = opaque(lane);
    const int TS = t_stride(a);
    constexpr int NG = (NA + 15) / 16;
    double areg[K][NG];
#pragma unroll
    for (int k = 0; k < K; ++k) {
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int e = 16 * g + (lane & 15);
        areg[k][g] = L.apack[k * NA + (e < NA ? e : NA - 1)];
      }
    }
    for (int t0 = 0; t0 < a.T; t0 += kEmThreads) {
      const int t = t0 + tid;
      const bool ok = t < a.T;
      const int tt = ok ? t : (a.T - 1);
      double re[D], im[D], q[K];
      load_frame(L, tt, re, im);
#pragma unroll
      for (int k = 0; k < K; ++k) q[k] = 0.0;
      static_for<0, D>([&](auto ic) {
        constexpr int i = ic;
        const double dg = re[i] * re[i] + im[i] * im[i];
#pragma unroll
        for (int k = 0; k < K; ++k) fmac_row_bcast<i % 16>(q[k], areg[k][i / 16], dg);
      });
      static_for<0, NOFF>([&](auto pc) {
        constexpr int p = pc;
        constexpr int i = tri_i<D>(p), j = tri_j<D>(p);
        constexpr int e0 = D + 2 * p, e1 = e0 + 1;
        const double pr = re[i] * re[j] + im[i] * im[j];
        const double pim = im[i] * re[j] - re[i] * im[j];
#pragma unroll
        for (int k = 0; k < K; ++k) {
          fmac_row_bcast<e0 % 16>(q[k], areg[k][e0 / 16], pr);
          fmac_row_bcast<e1 % 16>(q[k], areg[k][e1 / 16], pim);
        }
      });
      const double inv = L.inv_n2[tt];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const double qq = fmax(fabs(q[k] * inv), kTiny);
        if (ok) q_out[((size_t)b * K + k) * TS + tf + t] = qq;
      }
    }
  }

  // Spatial half of one iteration of the rotated joint loop (see JointMs).
  template <int MODE>
  static __device__ void run_joint_ms(const EmArgs& a, const JointMs& jm, char* smem) {
    static_assert(!SPILL, "joint kernels keep the observation in LDS");
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    if constexpr (MODE != 0) {
      if ((int)blockIdx.x >= jm.main_grid) {  // spectral finalize beside the bins
        spectral_finalize_block(jm.fin, (int)blockIdx.x - jm.main_grid, tid, kEmThreads,
                                reinterpret_cast<double*>(smem));
        return;
      }
    }
    const Lds L = carve(smem, a.T);
    for (int64_t b = blockIdx.x; b < a.B; b += jm.main_grid) {
      __syncthreads();
      if (tid < K) L.status[tid] = 0;
      if (tid == 0) *L.flags = 0;
      __syncthreads();
      if constexpr (MODE == 0) {
        phase_load(a, L, b, tid);
        __syncthreads();
        for (int k = wave; k < K; k += kEmWaves) prep_from_model(a, L, b, k, lane);
      } else {
        // observation, posteriors G and quadratic forms Q of a frame are requested together (one
        // memory round trip instead of two, one barrier less): the thread that stages frame t
        // also owns its M-step weights
        phase_load_gamma(a, L, b, tid, wave, lane);
        __syncthreads();
        if (jm.weight_fk_out && tid == 0) {
          double v[K], tot = 0.0;
#pragma unroll
          for (int k = 0; k < K; ++k) {
            v[k] = 0.0;
#pragma unroll
            for (int w = 0; w < kEmWaves; ++w) v[k] += L.red[w * K + k];
            tot += v[k];
          }
#pragma unroll
          for (int k = 0; k < K; ++k) jm.weight_fk_out[(size_t)b * K + k] = v[k] / tot;
        }
        switch (wave) {
          case 0: phase_m<0>(a, L, lane); break;
          case 1: phase_m<1>(a, L, lane); break;
          case 2: phase_m<2>(a, L, lane); break;
          default: phase_m<3>(a, L, lane); break;
        }
        __syncthreads();
        for (int k = wave; k < K; k += kEmWaves) factor_class(a, L, b, k, lane, MODE == 2);
      }
      __syncthreads();
      if constexpr (MODE != 2) {
        phase_q(a, L, b, tid, lane, jm.q_out);
        if (tid < K)
          jm.lndet_out[(size_t)b * K + tid] =
              log(L.detm[tid]) + (double)L.dete[tid] * 0.6931471805599453;
      }
      if (MODE != 0 && tid < K && a.out_status) a.out_status[(size_t)b * K + tid] = L.status[tid];
    }
  }

  // ===================== split-bin variant ====================================
  // G workgroups share ONE problem: each owns a window of `split_window` frames,
  // runs the E and M phases on it, and the partial covariance / class sums are
  // exchanged through an L2-resident slab once per iteration; every workgroup
  // then factors the (identical) totals redundantly.  Used for the few remainder
  // problems of a launch (B = 2^n + 1 frequency bins on a 256-CU device would
  // otherwise put a third full workgroup on one CU and set the kernel time).
  //
  // Hand-off protocol (cdna_hip_programming.md guideline 16, form R1): write-through
  // (sc1, relaxed agent-scope) slab stores -> every storing wave drains vmcnt(0) ->
  // __syncthreads -> one lane arrives on a monotonic agent-scope counter and polls it
  // (relaxed, bounded, s_sleep) -> __syncthreads -> sc1 loads of all slabs.  No cache
  // flushing fences (each costs ~1.7 us).  Slabs are double-buffered by iteration
  // parity; a timed-out spin sets *xerror and never hangs.
  static constexpr int kSlabLen = K * NA + K + 1;
  static constexpr unsigned kSpinLimit = 20000000u;  // ~2 s of polling before giving up

  // A bounded spin ran out (the peers of this group are not co-resident, e.g. under heavy
  // contention from other kernels): the sums of this problem are incomplete from here on.  The
  // per-launch word poisons the status output of every split problem of the launch (see the end
  // of run_split: order-independent atomic ORs onto status words zeroed before the launch), the
  // sticky word backs pbbss_split_error().
  static __device__ void split_timeout(const EmArgs& a) {
    atomicExch(a.xerror, a.xepoch ? a.xepoch : 1);
    atomicExch(a.xerror + 16, 1);
  }
  // did a bounded spin of THIS launch run out?  (launches stamp the word with their epoch, so it
  // never has to be cleared between launches; xepoch == 0: the launcher zeroes it)
  static __device__ bool split_failed(const EmArgs& a) {
    const int v = __hip_atomic_load(a.xerror, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return a.xepoch ? (v == a.xepoch) : (v != 0);
  }

  static __device__ void split_exchange(const EmArgs& a, const Lds& L, int prob, int nprob, int g,
                                        int it, int tid) {
    const int G = a.split_groups;
    double* slabs = a.xslab + ((size_t)(it & 1) * nprob + prob) * (size_t)G * kSlabLen;
    double* mine = slabs + (size_t)g * kSlabLen;
    PBBSS_DEV_ASSERT(a.xbuf_given == 0 ||
                     (size_t)(reinterpret_cast<char*>(slabs + (size_t)G * kSlabLen) -
                              reinterpret_cast<char*>(a.xcount)) <= a.xbuf_given);
    for (int idx = tid; idx < kSlabLen; idx += kEmThreads) {
      double v;
      if (idx < K * NA) {
        v = L.cpack[idx];  // slab order = packed order
      } else if (idx < K * NA + K) {
        const int k = idx - K * NA;
        v = 0.0;
#pragma unroll
        for (int w = 0; w < kEmWaves; ++w) v += L.red[w * K + k];
      } else {
        v = (double)(*L.flags & 1);
      }
      // write-through (sc1) store: visible in L2 without a release fence
      __hip_atomic_store(mine + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its stores
    __syncthreads();
    if (tid == 0) {
      unsigned* cnt = a.xcount + prob;
      __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)G * (unsigned)(it + 1);
      unsigned spins = 0;
      while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins >= (a.spin_limit ? a.spin_limit : kSpinLimit)) {
          split_timeout(a);
          break;
        }
      }
    }
    __syncthreads();
    // class sums first into registers (L.red is both source above and target below)
    for (int idx = tid; idx < kSlabLen; idx += kEmThreads) {
      // sc1 loads (L2, bypassing this CU's L1) pair with the sc1 stores: no acquire fence.
      // All peers' values are requested back-to-back (one L2 round trip, not G of them).
      double tot = 0.0;
      for (int g0 = 0; g0 < G; g0 += 8) {
        double part[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int gg = (g0 + u < G) ? g0 + u : g;  // clamp to a valid slab, masked below
          part[u] = __hip_atomic_load(slabs + (size_t)gg * kSlabLen + idx, __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) tot += (g0 + u < G) ? part[u] : 0.0;
      }
      if (idx < K * NA) {
        L.cpack[idx] = tot;
      } else if (idx < K * NA + K) {
        const int k = idx - K * NA;
        L.red[k] = tot;  // wave 0's slot carries the total, the others are cleared
#pragma unroll
        for (int w = 1; w < kEmWaves; ++w) L.red[w * K + k] = 0.0;
      } else {
        if (tot > 0.0) *L.flags |= 1;
      }
    }
    __syncthreads();
  }

  // ---- remainder problems of the one-iteration joint launch ---------------------------------
  // 513 bins on 256 CUs put a third full workgroup on one CU, and a one-iteration launch pays
  // that tail EVERY iteration (34 us per launch against ~20 us for a workgroup's own work).
  // The r remainder problems are therefore cut into G frame windows handled by small member
  // workgroups in the same grid (blocks >= main_grid): window load, E-step (affiliations and
  // q of the window go to HBM as usual), partial covariance sums -> an L2 slab (sc1 stores,
  // the protocol of split_exchange), then ONE arrival on a counter.  Nobody waits: the member
  // that arrives last sums the G slabs in a fixed order, factors the classes and writes the
  // state; the others are done.  No co-residency requirement, no spin, and the result does not
  // depend on which member happens to be last.  The last member also resets the counter.
  static constexpr int kJointCounterBase = 24;  // xcount[24 + prob]; [0, 24) belong to run_split

  static __device__ void run_joint_member(const EmArgs& ga, const JointExtras& jx0, char* smem) {
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int G = ga.split_groups;
    const int m = (int)blockIdx.x - jx0.main_grid;
    const int prob = m / G, g = m % G;
    const int64_t b = ga.b_first + prob;
    const int tf = g * ga.split_window;
    EmArgs a = ga;  // this workgroup's window
    a.T = min(ga.split_window, ga.T_total - tf);
    const Lds L = carve(smem, ga.split_window, nullptr);
    int* lastflag = reinterpret_cast<int*>(smem + lds_bytes(ga.split_window));
    if (tid < K) L.status[tid] = 0;
    if (tid == 0) *L.flags = 0;
    __syncthreads();
    phase_load(a, L, b, tid, tf);
    __syncthreads();
    for (int k = wave; k < K; k += kEmWaves) {
      if (jx0.state_in) {
        const double* st = jx0.state_in + ((size_t)b * K + k) * (NA + 2);
        for (int i = lane; i < NA; i += kWave) L.apack[k * NA + i] = st[i];
        if (lane == 0) {
          const double dm = st[NA];
          L.detm[k] = dm;
          L.rdet[k] = 1.0 / dm;
          L.dete[k] = (int)st[NA + 1];
        }
      } else {
        prep_from_model(a, L, b, k, lane);
      }
    }
    __syncthreads();
    JointExtras jx = jx0;
    jx.perm = nullptr;
    const bool owns = (wave << 6) < a.T;  // windows are <= 256 frames: one E pass
    if (a.iterations == 0) {
      if (owns) phase_e<true, true, true>(a, L, b, tid, wave, lane, a.final_eps, tf, &jx);
      return;
    }
    if (owns) {
      phase_e<false, true, true>(a, L, b, tid, wave, lane, a.aff_eps, tf, &jx);
    } else if (lane < K) {
      L.red[wave * K + lane] = 0.0;
    }
    __syncthreads();
    switch (wave) {
      case 0: phase_m<0>(a, L, lane); break;
      case 1: phase_m<1>(a, L, lane); break;
      case 2: phase_m<2>(a, L, lane); break;
      default: phase_m<3>(a, L, lane); break;
    }
    __syncthreads();
    double* slabs = ga.xslab + (size_t)prob * G * kSlabLen;
    double* mine = slabs + (size_t)g * kSlabLen;
    for (int idx = tid; idx < kSlabLen; idx += kEmThreads) {
      double v;
      if (idx < K * NA) {
        v = L.cpack[idx];
      } else if (idx < K * NA + K) {
        const int k = idx - K * NA;
        v = 0.0;
#pragma unroll
        for (int w = 0; w < kEmWaves; ++w) v += L.red[w * K + k];
      } else {
        v = (double)(*L.flags & 1);
      }
      __hip_atomic_store(mine + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its stores
    __syncthreads();
    if (tid == 0) {
      unsigned* cnt = ga.xcount + kJointCounterBase + prob;
      const unsigned before =
          __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = (before == (unsigned)(G - 1));
      if (last) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *lastflag = last;
    }
    __syncthreads();
    if (!*lastflag) return;
    for (int idx = tid; idx < kSlabLen; idx += kEmThreads) {
      double tot = 0.0;
      for (int g0 = 0; g0 < G; g0 += 8) {
        double part[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int gg = (g0 + u < G) ? g0 + u : g;
          part[u] = __hip_atomic_load(slabs + (size_t)gg * kSlabLen + idx, __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) tot += (g0 + u < G) ? part[u] : 0.0;
      }
      if (idx < K * NA) {
        L.cpack[idx] = tot;
      } else if (idx < K * NA + K) {
        const int k = idx - K * NA;
        L.red[k] = tot;
#pragma unroll
        for (int w = 1; w < kEmWaves; ++w) L.red[w * K + k] = 0.0;
      } else {
        if (tot > 0.0) *L.flags |= 1;
      }
    }
    __syncthreads();
    if (jx0.weight_fk_out && tid == 0) {
      double v[K], tot = 0.0;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        v[k] = L.red[k];
        tot += v[k];
      }
#pragma unroll
      for (int k = 0; k < K; ++k) jx0.weight_fk_out[(size_t)b * K + k] = v[k] / tot;
    }
    for (int k = wave; k < K; k += kEmWaves) {
      factor_class(a, L, b, k, lane, jx0.emit_model != 0);
      if (jx0.state_out) {
        double* st = jx0.state_out + ((size_t)b * K + k) * (NA + 2);
        for (int i = lane; i < NA; i += kWave) st[i] = L.apack[k * NA + i];
        if (lane == 0) {
          st[NA] = L.detm[k];
          st[NA + 1] = (double)L.dete[k];
        }
      }
    }
    __syncthreads();
    if (tid < K && a.out_status) a.out_status[(size_t)b * K + tid] = L.status[tid];
  }

  // Distributed factorisation of a split problem: after the exchange every member holds the
  // same totals, but instead of all G members factoring all K classes (K waves x ~500
  // instructions on CUs that already host two full workgroups) member k factors class k
  // alone and publishes the packed inverse, determinant, weight and status through a second
  // L2 slab; the others only read K * (NA + 5) doubles.  Same hand-off protocol as above,
  // second counter at xcount[16 + prob], double-buffered by iteration parity (a member can
  // only reach the same parity again after two further full exchanges, i.e. after every
  // member has read this one).
  static constexpr int kSlab2Len = NA + 5;
  static __host__ __device__ size_t split_slab_doubles(int nprob, int G) {
    return (size_t)2 * nprob * G * kSlabLen + (size_t)2 * nprob * K * kSlab2Len;
  }

  static __device__ void split_publish_model(const EmArgs& a, const Lds& L, int prob, int nprob,
                                             int g, int it, int tid) {
    const int G = a.split_groups;
    double* base = a.xslab + (size_t)2 * nprob * G * kSlabLen;
    double* slabs = base + ((size_t)(it & 1) * nprob + prob) * (size_t)K * kSlab2Len;
    if (g < K) {
      double* mine = slabs + (size_t)g * kSlab2Len;
      for (int idx = tid; idx < kSlab2Len; idx += kEmThreads) {
        double v;
        if (idx < NA) {
          v = L.apack[g * NA + idx];
        } else if (idx == NA) {
          v = L.detm[g];
        } else if (idx == NA + 1) {
          v = L.rdet[g];
        } else if (idx == NA + 2) {
          v = (double)L.dete[g];
        } else if (idx == NA + 3) {
          v = L.wgt[g];
        } else {
          v = (double)L.status[g];
        }
        __hip_atomic_store(mine + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (tid == 0) {
      unsigned* cnt = a.xcount + 16 + prob;
      if (g < K) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)K * (unsigned)(it + 1);
      unsigned spins = 0;
      while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins >= (a.spin_limit ? a.spin_limit : kSpinLimit)) {
          split_timeout(a);
          break;
        }
      }
    }
    __syncthreads();
    for (int idx = tid; idx < K * kSlab2Len; idx += kEmThreads) {
      const int k = idx / kSlab2Len, e = idx - k * kSlab2Len;
      if (k == g) continue;  // this member's own class is already in LDS
      const double v =
          __hip_atomic_load(slabs + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (e < NA) {
        L.apack[k * NA + e] = v;
      } else if (e == NA) {
        L.detm[k] = v;
      } else if (e == NA + 1) {
        L.rdet[k] = v;
      } else if (e == NA + 2) {
        L.dete[k] = (int)v;
      } else if (e == NA + 3) {
        L.wgt[k] = v;
      } else {
        L.status[k] = (int)v;  // cumulative in the publisher
      }
    }
    __syncthreads();
  }

  // mblock / nblocks: index of this member workgroup among the member workgroups of the launch
  // (the whole grid of the stand-alone split kernel; the blocks behind main_grid of the EM kernel)
  static __device__ void run_split(const EmArgs& ga, char* smem, int mblock, int nblocks) {
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int G = ga.split_groups;
    const int prob = mblock / G, g = mblock % G;
    const int nprob = nblocks / G;
    const int64_t b = ga.b_first + prob;
    const int tf = g * ga.split_window;
    // The split waves sit on CUs that also host two full workgroups; their work is a
    // sliver of the CU's but their dependency chain sets the launch time: let them win
    // the issue arbitration.
    switch (ga.split_prio) {
      case 1: __builtin_amdgcn_s_setprio(1); break;
      case 2: __builtin_amdgcn_s_setprio(2); break;
      case 3: __builtin_amdgcn_s_setprio(3); break;
      default: break;
    }
    EmArgs a = ga;  // this workgroup's window
    a.T = min(ga.split_window, ga.T_total - tf);
    const Lds L = carve(smem, ga.split_window, nullptr);
#ifdef PBBSS_PHASE_PROFILE
    unsigned long long spc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stprev = __builtin_readcyclecounter();
#define PBBSS_STICK(i)                                       \
  {                                                          \
    unsigned long long tn = __builtin_readcyclecounter();    \
    spc[i] += tn - stprev;                                   \
    stprev = tn;                                             \
  }
#else
#define PBBSS_STICK(i)
#endif
    if (tid < K) {
      L.status[tid] = 0;
      // the members OR their bits into the status words at the end (order-independent); member 0
      // zeroes them first -- a member that OR-ed a time-out in before this store is not lost:
      // member 0 sees the same stamped error word at its own end and ORs the poison in again
      if (g == 0 && a.out_status) a.out_status[(size_t)b * K + tid] = 0;
    }
    if (tid == 0) *L.flags = 0;
    __syncthreads();
    phase_load(a, L, b, tid, tf);
    __syncthreads();
    const bool model_in = (a.gamma0 == nullptr);
    if (model_in) {
      for (int k = wave; k < K; k += kEmWaves) prep_from_model(a, L, b, k, lane);
    } else {
      phase_init_gamma(a, L, b, tid, wave, lane, tf);
    }
    __syncthreads();
    for (int it = 0; it < a.iterations; ++it) {
      if (it > 0 || model_in) {
        // windows are <= 256 frames (one E pass): waves that own no frame skip the phase
        if ((wave << 6) < a.T) {
          phase_e<false, false>(a, L, b, tid, wave, lane, a.aff_eps, tf);
        } else if (lane < K) {
          L.red[wave * K + lane] = 0.0;
        }
        PBBSS_STICK(1)
        __syncthreads();
        PBBSS_STICK(2)
      }
      switch (wave) {
        case 0: phase_m<0>(a, L, lane); break;
        case 1: phase_m<1>(a, L, lane); break;
        case 2: phase_m<2>(a, L, lane); break;
        default: phase_m<3>(a, L, lane); break;
      }
      PBBSS_STICK(3)
      __syncthreads();
      split_exchange(a, L, prob, nprob, g, it, tid);
      PBBSS_STICK(4)
      const bool last = (it == a.iterations - 1);
      if (G >= K) {
        // member k factors class k (and writes that class of the model), then publishes it.
        // On the LAST wave: with 64-frame windows the E phase runs on wave 0 alone, and the
        // co-resident full workgroups are slowed by the most loaded SIMD, not by the total.
        if (g < K && wave == kEmWaves - 1) factor_class(a, L, b, g, lane, last);
        PBBSS_STICK(5)
        __syncthreads();
        split_publish_model(a, L, prob, nprob, g, it, tid);
        PBBSS_STICK(6)
      } else {
        // every workgroup factors the same totals; only group 0 writes the model
        EmArgs fa = a;
        if (g != 0) {
          fa.out_eigvec = nullptr;
          fa.out_eigval = nullptr;
          fa.out_cov = nullptr;
        }
        for (int k = wave; k < K; k += kEmWaves) factor_class(fa, L, b, k, lane, last);
        PBBSS_STICK(5)
        __syncthreads();
        PBBSS_STICK(6)
      }
    }
#ifdef PBBSS_PHASE_PROFILE
    // split launches report into the second half of the counter buffer: [32 + wave*8 + i]
    if (ga.prof && lane == 0) {
      for (int i = 0; i < 8; ++i) atomicAdd(ga.prof + 32 + wave * 8 + i, spc[i]);
    }
#endif
    if (tid < K) {
      if (g == 0 && a.out_weight && a.iterations > 0) a.out_weight[(size_t)b * K + tid] = L.wgt[tid];
      // A timed-out inter-workgroup wait invalidates the result: EVERY member that sees the
      // per-launch error word reports the solve as failed (NONFINITE -> the host raises), with
      // atomic ORs onto status words the launcher zeroed, so no ordering between the members'
      // exits is assumed.  Member 0 also contributes the regular status bits.
      const int bits = (g == 0 ? L.status[tid] : 0) |
                       (split_failed(a) ? (PBBSS_ST_EIG_NOCONV | PBBSS_ST_NONFINITE) : 0);
      if (a.out_status && bits) atomicOr(a.out_status + (size_t)b * K + tid, bits);
    }
    if (a.final_predict) phase_e<true, false>(a, L, b, tid, wave, lane, a.final_eps, tf);
    // Every member is past its last wait on the two arrival counters of this problem: the member
    // that leaves last puts them back to zero, so the next launch needs no memset in front of it
    // (xcount[8 + prob] counts the leavers).
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

  // ======== mixture weights shared by a group of problems =====================================
  // weight_constant_axis=(-3,) / (-3, -1) of the reference (cacgmm.py:59,
  // mixture_model_utils.py:184-201): the weights are averaged over the frequency bins of one
  // utterance, which couples the otherwise independent problems once per EM iteration.  All
  // `wgroup` problems of a group run as co-resident workgroups (the launcher guarantees it) and
  // exchange through agent-coherent (sc1) memory with a split-phase barrier: a workgroup
  // posts its masked affiliations (or their class sums) right after the E-step, runs its own
  // M-step and factorisation, and only then needs everybody else's contribution.
  //   SHARED_K  (-3, -1): every workgroup sums the B x K class sums itself (fixed order).
  //   SHARED_KT (-3,)   : 8-frame slices of the (K, T) weight plane are reduced by the first
  //                       ceil(T / 8) workgroups of the group and re-published.
  // Counters are monotonic: gcount[16 g] posts, gcount[16 g + 8] reduced slices.
  static constexpr int kSliceFrames = 8;

  // The cooperative kernel needs EVERY workgroup of a group co-resident; other kernels of the
  // process competing for the compute units can keep that from happening for as long as they
  // run (tests/test_gpu_contention.py).  The wait is therefore short (~1 s: a healthy barrier
  // takes microseconds) and sticky -- once one workgroup has given up, every later wait of the
  // launch falls through at once -- and the host layer repeats a fit that was reported this way
  // on the step-wise path, which needs no co-residency.
  static constexpr unsigned kSharedSpinLimit = 3000000u;
  static __device__ void shared_spin(const EmArgs& a, unsigned* cnt, unsigned target) {
    unsigned spins = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if ((spins & 4095u) == 0u && split_failed(a)) break;
      if (++spins >= (a.spin_limit ? a.spin_limit : kSharedSpinLimit)) {
        split_timeout(a);
        break;
      }
    }
  }

  // after the barrier that follows an E-step (or the initialisation): publish, count
  static __device__ void shared_post(const EmArgs& a, const Lds& L, int64_t b, int step, int tid) {
    if (a.weight_mode == PBBSS_WEIGHT_SHARED_K && tid < K) {
      double v = 0.0;
#pragma unroll
      for (int w = 0; w < kEmWaves; ++w) v += L.red[w * K + tid];
      __hip_atomic_store(a.gsum + ((size_t)(step & 1) * a.B + b) * K + tid, v, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // SHARED_KT: the E-step stored the affiliations and every wave drained them before the barrier
    if (tid == 0)
      __hip_atomic_fetch_add(a.gcount + (b / a.wgroup) * 16, 1u, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
  }

  // SHARED_K: weights of `step` -> L.wgt
  static __device__ void shared_acquire_k(const EmArgs& a, const Lds& L, int64_t b, int step,
                                          int tid, int wave, int lane) {
    const int64_t grp = b / a.wgroup;
    if (tid == 0) shared_spin(a, a.gcount + grp * 16, (unsigned)a.wgroup * (unsigned)(step + 1));
    __syncthreads();
    const double* src = a.gsum + ((size_t)(step & 1) * a.B + grp * a.wgroup) * K;
    double acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = 0.0;
    for (int i = tid; i < a.wgroup; i += kEmThreads) {
#pragma unroll
      for (int k = 0; k < K; ++k)
        acc[k] += __hip_atomic_load(src + (size_t)i * K + k, __ATOMIC_RELAXED,
                                    __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      double tot = wave_sum(acc[k]);
      if (lane == 0) L.red[wave * K + k] = tot;
    }
    __syncthreads();
    if (tid < K) {
      double mine = 0.0, norm = 0.0;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        double tk = 0.0;
#pragma unroll
        for (int w = 0; w < kEmWaves; ++w) tk += L.red[w * K + k];
        norm += fabs(tk);
        if (k == tid) mine = tk;
      }
      if (a.saliency) {
        L.wgt[tid] = mine / ((norm == 0.0) ? 1e-10 : norm);  // mixture_model_utils.py:190-201
      } else {
        L.wgt[tid] = mine / ((double)a.wgroup * (double)t_stride(a));  // :186-188
      }
    }
    __syncthreads();
  }

  // SHARED_KT: reduce the slices owned by this workgroup (r = index within the group)
  static __device__ void shared_reduce_kt(const EmArgs& a, const Lds& L, int64_t b, int step,
                                          int tid, int wave, int lane) {
    const int TS = t_stride(a);
    const int64_t grp = b / a.wgroup;
    const int r = (int)(b - grp * a.wgroup);
    const int nsl = (TS + kSliceFrames - 1) / kSliceFrames;
    if (r >= nsl) return;
    if (tid == 0) shared_spin(a, a.gcount + grp * 16, (unsigned)a.wgroup * (unsigned)(step + 1));
    __syncthreads();
    const double* src = a.gaff + ((size_t)(step & 1) * a.B + grp * a.wgroup) * K * TS;
    double* dst = a.gw + ((size_t)(step & 1) * (a.B / a.wgroup) + grp) * K * TS;
    const int j = tid & (kSliceFrames - 1);
    const int part = tid / kSliceFrames;             // 32 parts over the problems of the group
    constexpr int kParts = kEmThreads / kSliceFrames;
    double* tmp = L.wbuf;                            // free between factorisation and the next E-step
    for (int sl = r; sl < nsl; sl += a.wgroup) {
      const int t = sl * kSliceFrames + j;
      const bool ok = t < TS;
      double acc[K];
#pragma unroll
      for (int k = 0; k < K; ++k) acc[k] = 0.0;
      for (int i0 = part; i0 < a.wgroup; i0 += 4 * kParts) {
        double v[4][K];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u * kParts;
          const bool in = ok && i < a.wgroup;
#pragma unroll
          for (int k = 0; k < K; ++k)
            v[u][k] = in ? __hip_atomic_load(src + ((size_t)i * K + k) * TS + t, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT)
                         : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int k = 0; k < K; ++k) acc[k] += v[u][k];
      }
      // lanes of one wave: 8 parts x 8 frames -> butterfly over the parts
#pragma unroll
      for (int k = 0; k < K; ++k) {
        acc[k] += __shfl_xor(acc[k], 8);
        acc[k] += __shfl_xor(acc[k], 16);
        acc[k] += __shfl_xor(acc[k], 32);
        if (lane < kSliceFrames) tmp[(wave * K + k) * kSliceFrames + lane] = acc[k];
      }
      __syncthreads();
      if (tid < kSliceFrames && ok) {
        double tk[K], norm = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          tk[k] = 0.0;
#pragma unroll
          for (int w = 0; w < kEmWaves; ++w) tk[k] += tmp[(w * K + k) * kSliceFrames + tid];
          norm += fabs(tk[k]);
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const double w = a.saliency ? tk[k] / ((norm == 0.0) ? 1e-10 : norm)  // :190-201
                                      : tk[k] / (double)a.wgroup;               // :186-188
          __hip_atomic_store(dst + (size_t)k * TS + t, w, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
      if (tid == 0)
        __hip_atomic_fetch_add(a.gcount + grp * 16 + 8, 1u, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
  }

  // SHARED_KT: wait until every slice of `step` is reduced; returns the (K, T) weight plane
  static __device__ const double* shared_acquire_kt(const EmArgs& a, int64_t b, int step, int tid) {
    const int TS = t_stride(a);
    const int64_t grp = b / a.wgroup;
    const int nsl = (TS + kSliceFrames - 1) / kSliceFrames;
    if (tid == 0) shared_spin(a, a.gcount + grp * 16 + 8, (unsigned)nsl * (unsigned)(step + 1));
    __syncthreads();
    return a.gw + ((size_t)(step & 1) * (a.B / a.wgroup) + grp) * K * TS;
  }

  static __device__ void run_shared(const EmArgs& a, char* smem) {
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const Lds L = carve(smem, a.T, nullptr);
    const int64_t b = a.b_first + blockIdx.x;  // one workgroup per problem, all co-resident
    const bool kt = (a.weight_mode == PBBSS_WEIGHT_SHARED_KT);
    const int TS = t_stride(a);
    if (tid < K) L.status[tid] = 0;
    if (tid == 0) *L.flags = 0;
    __syncthreads();
    phase_load(a, L, b, tid);
    __syncthreads();
    const bool model_in = (a.gamma0 == nullptr);
    if (model_in) {
      // the model of the caller carries one weight set per group: (B / wgroup, K[, T])
      for (int k = wave; k < K; k += kEmWaves) prep_from_model(a, L, b, k, lane);
      __syncthreads();
      if (!kt && tid < K) L.wgt[tid] = a.in_weight[(b / a.wgroup) * K + tid];
    } else {
      phase_init_gamma(a, L, b, tid, wave, lane);  // SHARED_KT: posts the masked initialisation
    }
    if (kt) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int it = 0; it < a.iterations; ++it) {
      if (it > 0 || model_in) {
        double* pub = kt ? a.gaff + ((size_t)(it & 1) * a.B + b) * K * TS : nullptr;
        if (kt) {
          // first E-step of a fit that starts from a model: the weight plane of the caller
          const double* w_kt = (it == 0) ? a.in_weight + (b / a.wgroup) * (int64_t)K * TS
                                         : shared_acquire_kt(a, b, it - 1, tid);
          phase_e<false, true, false, false>(a, L, b, tid, wave, lane, a.aff_eps, 0, nullptr, w_kt,
                                             pub);
        } else {
          if (it > 0) shared_acquire_k(a, L, b, it - 1, tid, wave, lane);
          phase_e<false, false, false, PBBSS_E_PAIR != 0>(a, L, b, tid, wave, lane, a.aff_eps);
        }
        if (kt) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
      shared_post(a, L, b, it, tid);
      switch (wave) {
        case 0: phase_m<0>(a, L, lane); break;
        case 1: phase_m<1>(a, L, lane); break;
        case 2: phase_m<2>(a, L, lane); break;
        default: phase_m<3>(a, L, lane); break;
      }
      __syncthreads();
      const bool last = (it == a.iterations - 1);
      for (int k = wave; k < K; k += kEmWaves) factor_class(a, L, b, k, lane, last);
      __syncthreads();
      if (kt) shared_reduce_kt(a, L, b, it, tid, wave, lane);
    }
    const double* w_kt = nullptr;
    if (a.iterations > 0) {
      const int64_t grp = b / a.wgroup;
      if (kt) {
        w_kt = shared_acquire_kt(a, b, a.iterations - 1, tid);
        if (a.out_weight_shared && b == grp * a.wgroup) {
          for (int i = tid; i < K * TS; i += kEmThreads)
            a.out_weight_shared[(size_t)grp * K * TS + i] =
                __hip_atomic_load(w_kt + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      } else {
        shared_acquire_k(a, L, b, a.iterations - 1, tid, wave, lane);
        if (a.out_weight_shared && b == grp * a.wgroup && tid < K)
          a.out_weight_shared[(size_t)grp * K + tid] = L.wgt[tid];
      }
    }
    if (tid < K && a.out_status) {
      int st = L.status[tid];
      if (split_failed(a))
        st |= PBBSS_ST_EIG_NOCONV | PBBSS_ST_NONFINITE;  // a hand-off timed out: results are void
      a.out_status[(size_t)b * K + tid] = st;
    }
    if (a.final_predict) {
      if (w_kt) {
        phase_e<true, true>(a, L, b, tid, wave, lane, a.final_eps, 0, nullptr, w_kt);
      } else {
        phase_e<true, false>(a, L, b, tid, wave, lane, a.final_eps);
      }
    }
  }

  static __device__ void run(const EmArgs& a, char* smem) {
    const int tid = threadIdx.x;
    // wave index is uniform across the wavefront: tell the compiler so the phase
    // dispatch below is a scalar branch, not four exec-masked code paths
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    // (The split groups of the remainder problems are a SECOND kernel on a side stream, not blocks
    // of this grid: with run_split compiled into this function -- inlined or called -- hipcc's
    // register allocation of the main loop degrades from 9 to 76 spilled VGPRs, 1.475 -> 1.65 ms
    // per fit on one box, profiles/r03_a_member_modes.txt.)
    PBBSS_DEV_ASSERT(a.lds_given == 0 || lds_bytes(a.T) <= a.lds_given);
    const Lds L = carve(smem, a.T, SPILL ? a.scratch + (size_t)blockIdx.x * a.scratch_stride : nullptr);
#ifdef PBBSS_PHASE_PROFILE
    unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_readcyclecounter();
#define PBBSS_TICK(i)                                        \
  {                                                          \
    unsigned long long tn = __builtin_readcyclecounter();    \
    pc[i] += tn - tprev;                                     \
    tprev = tn;                                              \
  }
#else
#define PBBSS_TICK(i)
#endif
    for (int64_t b = blockIdx.x; b < a.B; b += gridDim.x) {
      __syncthreads();  // previous problem fully retired before LDS is reused
      if (tid < K) L.status[tid] = 0;
      if (tid == 0) *L.flags = 0;
      __syncthreads();
      phase_load(a, L, b, tid);
      __syncthreads();
      const bool model_in = (a.gamma0 == nullptr);
      if (model_in) {
        for (int k = wave; k < K; k += kEmWaves) prep_from_model(a, L, b, k, lane);
      } else {
        phase_init_gamma(a, L, b, tid, wave, lane);
      }
      __syncthreads();
      PBBSS_TICK(0)
      for (int it = 0; it < a.iterations; ++it) {
        if (it > 0 || model_in) {
          phase_e<false, false, false, PBBSS_E_PAIR != 0>(a, L, b, tid, wave, lane, a.aff_eps);
          PBBSS_TICK(1)
          __syncthreads();
          PBBSS_TICK(2)
        }
        switch (wave) {
          case 0: phase_m<0>(a, L, lane); break;
          case 1: phase_m<1>(a, L, lane); break;
          case 2: phase_m<2>(a, L, lane); break;
          default: phase_m<3>(a, L, lane); break;
        }
        PBBSS_TICK(3)
        __syncthreads();
        PBBSS_TICK(4)
        const bool last = (it == a.iterations - 1);
        for (int k = wave; k < K; k += kEmWaves) factor_class(a, L, b, k, lane, last);
        PBBSS_TICK(5)
        __syncthreads();
        PBBSS_TICK(6)
      }
      if (tid < K) {
        if (a.out_weight && a.iterations > 0) a.out_weight[(size_t)b * K + tid] = L.wgt[tid];
        if (a.out_status) a.out_status[(size_t)b * K + tid] = L.status[tid];
      }
      if (a.final_predict) {
        // after a fit: model.predict(y) with the new per-class weights; a bare
        // predict (iterations == 0) may carry frame-varying weights (wt != 0)
        if (a.iterations == 0 && a.wt != 0) {
          phase_e<true, true>(a, L, b, tid, wave, lane, a.final_eps);
        } else {
          phase_e<true, false>(a, L, b, tid, wave, lane, a.final_eps);
        }
      }
      PBBSS_TICK(7)
    }
#ifdef PBBSS_PHASE_PROFILE
    // [wave][8] cycle sums over all workgroups: 0 load/init, 1 E, 2 E-barrier+sums,
    // 3 M, 4 M-barrier, 5 factor, 6 factor-barrier, 7 final predict
    if (a.prof && lane == 0) {
      for (int i = 0; i < 8; ++i) atomicAdd(a.prof + wave * 8 + i, pc[i]);
    }
#endif
  }
};

// Register budget: K <= 4 fits 168 VGPRs (3 workgroups per CU, the LDS limit at T=500);
// more classes need more M-phase accumulators (16 float64 per class) -> 2 per CU.
#ifndef PBBSS_EM_WAVES
#define PBBSS_EM_WAVES 3
#endif
constexpr int em_waves_per_simd(int K) { return K <= 4 ? PBBSS_EM_WAVES : 2; }

template <int D, int K, typename YS, bool SPILL>
__global__ void __launch_bounds__(kEmThreads, em_waves_per_simd(K)) cacgmm_em_kernel(EmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  EmKernel<D, K, YS, SPILL>::run(a, smem);
}


// PA: with the per-bin permutation search of the inline aligner (its own instantiation: the
// search and the class selects of the E-step would otherwise cost the default path 224 bytes of
// scratch per lane -- profiles/r04_*_config5_profile.txt: 25 MB of spill writes per launch)
template <int D, int K, typename YS, bool PA>
__global__ void __launch_bounds__(kEmThreads, em_waves_per_simd(K))
    cacgmm_joint_kernel(EmArgs a, JointExtras jx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  EmKernel<D, K, YS, false>::template run_joint<PA>(a, jx, smem);
}

template <int D, int K, typename YS, int MODE>
__global__ void __launch_bounds__(kEmThreads, em_waves_per_simd(K))
    cacgmm_joint_ms_kernel(EmArgs a, JointMs jm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  EmKernel<D, K, YS, false>::template run_joint_ms<MODE>(a, jm, smem);
}

template <int D, int K, typename YS>
__global__ void __launch_bounds__(kEmThreads, em_waves_per_simd(K)) cacgmm_em_shared_kernel(EmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  EmKernel<D, K, YS, false>::run_shared(a, smem);
}

template <int D, int K, typename YS>
__global__ void __launch_bounds__(kEmThreads, em_waves_per_simd(K)) cacgmm_em_split_kernel(EmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  EmKernel<D, K, YS, false>::run_split(a, smem, (int)blockIdx.x, (int)gridDim.x);
}

}  // namespace pbbss
