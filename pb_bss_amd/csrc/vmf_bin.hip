// Persistent von-Mises-Fisher mixture EM for MANY SMALL mixtures, round-6 kernel
// (BASELINE configs[3], vMF leg: one mixture per frequency bin on the
// 2 D-dimensional real features of a D-sensor observation -- 257 mixtures of 800 rows, E = 12).
//
// Reference: distribution/vmfmm.py:124-172 (VMFMMTrainer._fit / _m_step), :57-78 (predict, unit
// rows), distribution/von_mises_fisher.py:62-78 (log_pdf), :119-144 (fit: mean direction, the
// concentration of Banerjee 2005 eq. 4.4, clipped), :33-44 (log-Bessel normaliser),
// distribution/mixture_model_utils.py:30-47 (posterior), :184-201 (weights, saliency form).
//
// Why a second kernel (the round-4 one, embed.hip: vmf_bin_em_kernel, stays as the route for
// every other shape): rocprofv3 counted 2 331 VALU instructions per wave and EM iteration against
// a floor of ~390 multiply-adds, and one wavefront per SIMD.  What the instructions were: a
// run-time feature count (no unrolling: one ds_read_b32 + cvt per element and class mean), |y|^2,
// a square root and a division per row and iteration although the rows never change, the M-step
// as a second sweep over the tile with a (slot, dimension) thread mapping (tile and weights read
// again, 21 slot partials per sum reduced through LDS), three IEEE divisions per row, four
// workgroup barriers per iteration.  Here
//   * the feature count is a template parameter (EP = E padded to a multiple of four, zeros in
//     the padding): a row is EP/4 ds_read_b128 from a 64-row chunk layout (the plane of a chunk
//     at a compile-time offset from the lane's address, cacgmm_em.hpp: Lds), held in registers
//     for BOTH halves of the iteration;
//   * the class means are DPP operands (one register per class whose 16-lane rows hold the EP <=
//     16 components, pbbss_dev.hpp: fmac_row_bcast): no LDS read per multiply-add;
//   * 1/|y_n| and the saliency are staged once;
//   * E and M are ONE sweep, lane = row: posterior -> weights -> K (EP + 1) per-lane accumulators
//     -> one halving butterfly per wave and iteration (wave_reduce_scatter), partials of the NW
//     waves added by the wave that owns the class; two barriers per iteration;
//   * the log-Bessel normaliser -- a serial chain on the one wavefront per class the workgroup
//     waits for -- evaluates a tabulated polynomial instead of anchoring every lane's block of
//     the series with two lgamma calls per iteration (wave_log_bessel_fixed below);
//   * NW = 8 wavefronts per mixture when it has more than four chunks and a compute unit of its
//     own: two per SIMD instead of one, the chunks of a mixture are independent until the butterfly.
#include "embed.hpp"
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include "pbbss_dev.hpp"
#include "embed_dev.hpp"

namespace pbbss {
namespace {

constexpr double kLn2PiB = 1.8378770664093454;  // ln(2 pi)

struct VmfBinArgs {
  const void* y;  // (B, N, E) rows, float or double
  int N, E, iterations;
  const double* gamma;      // (B,K,N) initialisation or null (predict: the model below)
  const double* sal;        // (B,N) or null
  const double* in_mean;    // (B,K,E)
  const double* in_conc;    // (B,K)
  const double* in_weight;  // (B,K)
  double cmin, cmax;
  int weight_mode;
  double* out_mean;
  double* out_conc;
  double* out_weight;
  double* out_aff;  // (B,K,N) or null
};

// sum over the 16 lanes of a row (every row holds the same 16 values): four DPP steps
__device__ __forceinline__ double row16_sum(double v) {
  v += dpp_f64<kDppQuadXor1, 0xF>(v, v);
  v += dpp_f64<kDppQuadXor2, 0xF>(v, v);
  v += dpp_f64<kDppRowHalfMirror, 0xF>(v, v);
  v += dpp_f64<kDppRowMirror, 0xF>(v, v);
  return v;
}

// ln(I_nu(x) / x^nu) by the ascending series with a FIXED block of R terms per lane (embed_dev.hpp:
// wave_log_bessel_over_power sizes the blocks by x and anchors every lane's block with two lgamma
// calls -- ~1 500 instructions with divergent argument ranges, on the one wavefront per class the
// whole workgroup waits for: half of the iteration).  The order nu and the block size do not
// change during a fit, so everything that does not depend on x is tabulated ONCE in LDS
// (bessel_table: kBesselRows rows of 64 lanes):
//   row 0      c0 = lgamma(m0 + 1) + lgamma(m0 + nu + 1),  m0 = R lane  (anchor: ln t_m0 = m0 ln q - c0)
//   row r >= 1 c_r = prod_{j <= r} 1 / ((m0 + j)(m0 + j + nu))          (t_(m0+r) / t_m0 = c_r q^r)
// with q = x^2 / 4.  An evaluation is then one logarithm for the anchor, the block's sum relative
// to its first term as a polynomial in q (Horner, R - 1 multiply-adds -- it was a chain of R - 1
// divisions), and one max / sum over the wave.  R = ceil((ceil(x_max) + 48) / 64) <= 16 covers
// every x <= x_max (terms fall by > 4x per step past m = x); q^(R-1) c_(R-1) cannot overflow
// (q < 2.4e5, 15 factors).  Lanes whose block lies in the tail contribute exp(-huge) = 0.
constexpr int kBesselRows = 16;
__device__ __forceinline__ void bessel_table(double* tab, double nu, int R, int lane) {
  const double m0 = (double)(lane * R);
  tab[lane] = lgamma(m0 + 1.0) + lgamma(m0 + nu + 1.0);
  double c = 1.0;
  for (int r = 1; r < kBesselRows; ++r) {
    const double m1 = m0 + (double)r;
    c = (r < R) ? c / (m1 * (m1 + nu)) : 0.0;
    tab[r * kWave + lane] = c;
  }
}
__device__ __forceinline__ double wave_log_bessel_fixed(const double* tab, double nu, double x,
                                                        int lane, int R) {
  double c[kBesselRows];
#pragma unroll
  for (int r = 0; r < kBesselRows; ++r) c[r] = tab[r * kWave + lane];  // in flight during the log
  const double q = 0.25 * x * x;
  // (log_pos / exp_nonpos, pbbss_dev.hpp: x is the clamped concentration -- finite, positive and
  // normal, or NaN, which both propagate)
  const double lx = 2.0 * log_pos(x) - 1.3862943611198906;  // ln(x^2 / 4)
  const int m0 = lane * R;
  const double lt = (m0 ? (double)m0 * lx : 0.0) - c[0];
  double s = 0.0;  // rows >= R hold zeros: the Horner chain may always start at the last row
  if (R > 9) {
#pragma unroll
    for (int r = kBesselRows - 1; r >= 9; --r) s = fma(s, q, c[r]);
  }
#pragma unroll
  for (int r = 8; r >= 1; --r) s = fma(s, q, c[r]);
  s = fma(s, q, 1.0);
  const double la = lt + log_pos(s);  // s >= 1
  const double gmx = wave_max(la);
  const double sum = wave_sum(exp_nonpos(la - gmx));  // >= 1: the maximal lane contributes 1
  return -nu * 0.6931471805599453 + gmx + log_pos(sum);
}

template <int K, int EP, typename TS, int NW>
struct VmfBin {
  static constexpr int VW = 16 / (int)sizeof(TS);  // elements per 16-byte LDS vector
  static constexpr int P = EP / VW;                // vectors ("planes") per row
  static constexpr int NSUM = K * EP + K;          // S1[k][e], then S0[k]
  static constexpr int NACC = ((NSUM + 15) / 16) * 16;
  static constexpr int R = NACC / 16;              // totals per lane after the butterfly
  using Vec = typename std::conditional<std::is_same<TS, float>::value, float4, double2>::type;
  static_assert(EP % 4 == 0 && EP >= 4 && EP <= 16, "one DPP operand register per class");
  static_assert(NACC <= 64, "accumulators live in registers");

  static __host__ __device__ int chunks(int N) { return (N + kWave - 1) / kWave; }
  static __host__ __device__ size_t lds_bytes(int N) {
    const size_t np = (size_t)chunks(N) * kWave;
    return np * EP * sizeof(TS) +
           (2 * np + (size_t)K * 16 + (size_t)NW * NACC + 3 * K + kBesselRows * kWave) * sizeof(double);
  }

  // row of chunk c held by `lane`, widened
  static __device__ __forceinline__ void load_row(const Vec* tile, int c, int lane, double (&v)[EP]) {
    const Vec* p = tile + (size_t)c * P * kWave + lane;
    static_for<0, P>([&](auto pc) {
      constexpr int pl = pc;
      const Vec x = p[pl * kWave];
      if constexpr (VW == 4) {
        v[4 * pl] = (double)x.x;
        v[4 * pl + 1] = (double)x.y;
        v[4 * pl + 2] = (double)x.z;
        v[4 * pl + 3] = (double)x.w;
      } else {
        v[2 * pl] = (double)x.x;
        v[2 * pl + 1] = (double)x.y;
      }
    });
  }

  // posterior of one row from the model (mixture_model_utils.py:30-47): m[k] = class mean as DPP
  // operand register, lp = kappa_k <mu_k, y / |y|> + offset_k (von_mises_fisher.py:71-77)
  static __device__ __forceinline__ void posterior(const double (&v)[EP], double inv,
                                                   const double (&m)[K], const double (&prec)[K],
                                                   const double (&off)[K], const double (&wgt)[K],
                                                   double (&g)[K]) {
    double lp[K], mx = -1.79e308;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      double dot = 0.0;
      static_for<0, EP>([&](auto ec) {
        constexpr int e = ec;
        fmac_row_bcast<e>(dot, m[k], v[e]);
      });
      lp[k] = fma(prec[k], dot * inv, off[k]);
      mx = fmax(mx, lp[k]);
    }
    double den = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      g[k] = exp_nonpos(lp[k] - mx) * wgt[k];
      den += g[k];
    }
    // np.maximum keeps a NaN, v_max drops it: the NaN of a poisoned class sum rides on the
    // reciprocal instead (den - den is 0 for a finite sum)
    const double rden = fast_rcp(fmax(den, kTiny)) + (den - den);
#pragma unroll
    for (int k = 0; k < K; ++k) g[k] *= rden;
  }

  static __device__ void run(const VmfBinArgs& a, char* smraw) {
    const int N = a.N, E = a.E;
    const int nchunk = chunks(N), np = nchunk * kWave;
    Vec* tile = reinterpret_cast<Vec*>(smraw);                          // [chunk][P][64] x 16 B
    double* rinv = reinterpret_cast<double*>(smraw + (size_t)np * EP * sizeof(TS));  // [np] 1/|y_n|
    double* sv = rinv + np;                                             // [np] saliency, 0 = padding
    double* mu = sv + np;                                               // [K][16]
    double* red = mu + K * 16;                                          // [NW][NACC]
    double* sprec = red + NW * NACC;                                    // [K]
    double* soff = sprec + K;                                           // [K]
    double* swgt = soff + K;                                            // [K]
    double* btab = swgt + K;                                            // [kBesselRows][64]
    const int64_t b = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const TS* base = static_cast<const TS*>(a.y) + (size_t)b * N * E;
    {  // stage the rows into the chunk layout, zeros in the padding (rows >= N, features >= E);
       // unconditional loads at clamped indices, 8 in flight per thread
      TS* flat = reinterpret_cast<TS*>(smraw);
      constexpr int U = 8;
      const int total = np * EP;
      const int last = N * E - 1;
      for (int i0 = tid; i0 < total; i0 += U * NW * kWave) {
        TS raw[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int i = i0 + u * NW * kWave;
          const int n = i / EP, e = i - n * EP;
          const int src = n * E + e;
          raw[u] = base[(n < N && e < E && i < total) ? src : last];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int i = i0 + u * NW * kWave;
          if (i < total) {
            const int n = i / EP, e = i - n * EP;
            const int c = n >> 6, l = n & 63;
            flat[((size_t)(c * P + e / VW) * kWave + l) * VW + (e % VW)] =
                (n < N && e < E) ? raw[u] : (TS)0;
          }
        }
      }
    }
    if (!a.gamma) {  // predict only (iterations == 0): the model comes from the caller
      for (int i = tid; i < K * 16; i += NW * kWave) {
        const int k = i >> 4, e = i & 15;
        mu[i] = (e < E) ? a.in_mean[((size_t)b * K + k) * E + e] : 0.0;
      }
      for (int k = wave; k < K; k += NW) {
        const double conc = a.in_conc[b * K + k];
        const double o = -(0.5 * E * kLn2PiB + wave_log_bessel_over_power(0.5 * E - 1.0, conc, lane));
        if (lane == 0) {
          sprec[k] = conc;
          soff[k] = o;
          swgt[k] = a.in_weight[b * K + k];
        }
      }
    }
    __syncthreads();
    for (int n = tid; n < np; n += NW * kWave) {  // 1/|y_n| (unit rows, vmfmm.py:76-78), saliency
      double v[EP];
      load_row(tile, n >> 6, n & 63, v);
      double n2 = 0.0;
#pragma unroll
      for (int e = 0; e < EP; ++e) n2 = fma(v[e], v[e], n2);
      rinv[n] = 1.0 / fmax(sqrt(n2), kTiny);
      sv[n] = (n < N) ? (a.sal ? a.sal[(size_t)b * N + n] : 1.0) : 0.0;  // vmfmm.py:167
    }
    __syncthreads();
    const int nact = nchunk < NW ? nchunk : NW;  // waves that own at least one chunk
    // log-Bessel normaliser: fixed blocks while the clamp bounds the concentration (the default
    // max_concentration is 500: R = 9), else the x-sized blocks of embed_dev.hpp
    const double nu = 0.5 * E - 1.0;
    const int bR = (a.cmax <= 960.0) ? ((int)ceil(fmax(a.cmax, 1.0)) + 48 + kWave - 1) / kWave : 0;
    if (bR > 0 && a.iterations > 0 && wave == NW - 1) bessel_table(btab, nu, bR, lane);
    // (visible to the class waves after the first iteration's barrier)
    for (int it = 0; it < a.iterations; ++it) {
      if (wave < nact) {
        double acc[NACC];
#pragma unroll
        for (int x = 0; x < NACC; ++x) acc[x] = 0.0;
        const bool from_gamma = (it == 0);  // iterations > 0 start from affiliations
        double m[K], prec[K], off[K], wgt[K];
        if (!from_gamma) {
#pragma unroll
          for (int k = 0; k < K; ++k) {
            m[k] = mu[k * 16 + (lane & 15)];
            prec[k] = sprec[k];
            off[k] = soff[k];
            wgt[k] = swgt[k];
          }
        }
        for (int c = wave; c < nchunk; c += NW) {
          const int n = c * kWave + lane;
          double v[EP], g[K];
          load_row(tile, c, lane, v);
          const double inv = rinv[n];
          const double s = sv[n];
          if (from_gamma) {
            const int nc = n < N ? n : N - 1;
#pragma unroll
            for (int k = 0; k < K; ++k) g[k] = a.gamma[((size_t)b * K + k) * N + nc];
          } else {
            posterior(v, inv, m, prec, off, wgt, g);
          }
#pragma unroll
          for (int k = 0; k < K; ++k) {
            const double wk = g[k] * s;
            acc[K * EP + k] += wk;
            const double aw = wk * inv;
#pragma unroll
            for (int e = 0; e < EP; ++e) acc[k * EP + e] = fma(aw, v[e], acc[k * EP + e]);
          }
        }
        wave_reduce_scatter<NACC>(acc, lane);
        if ((lane & 3) == 0) {
          double* dst = red + wave * NACC + reduce_scatter_base<NACC>(lane);
#pragma unroll
          for (int x = 0; x < R; ++x) dst[x] = acc[x];
        }
      }
      __syncthreads();
      // ---- model: one wavefront per class (von_mises_fisher.py:122-144)
      for (int k = wave; k < K; k += NW) {
        const int e = lane & 15;
        // partial sums of the waves, ascending; all loads issued up front (NW is static, a wave
        // without chunks left its slots untouched: masked)
        double tw[NW], sw[NW][K];
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          const int wc = w < nact ? w : 0;
          tw[w] = red[wc * NACC + k * EP + (e < EP ? e : 0)];
#pragma unroll
          for (int j = 0; j < K; ++j) sw[w][j] = red[wc * NACC + K * EP + j];
        }
        double t = 0.0, s0[K];
#pragma unroll
        for (int j = 0; j < K; ++j) s0[j] = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          if (w < nact) {
            t += tw[w];
#pragma unroll
            for (int j = 0; j < K; ++j) s0[j] += sw[w][j];
          }
        }
        t = (e < EP) ? t : 0.0;
        const double n2 = row16_sum(t * t);
        // |r| and 1 / max(|r|, tiny) from one reciprocal square root (eq. 2.4); n2 = 0 or
        // denormal: the IEEE route
        double norm, rn;
        if (n2 > 1e-290 && n2 < 1e290) {
          rn = fast_rsqrt(n2);
          norm = n2 * rn;
        } else {
          norm = sqrt(n2);
          rn = 1.0 / fmax(norm, kTiny);
        }
        if (lane < 16) mu[k * 16 + lane] = t * rn;
        double s0k = 0.0, tabs = 0.0;
#pragma unroll
        for (int j = 0; j < K; ++j) {
          s0k = (j == k) ? s0[j] : s0k;
          tabs += fabs(s0[j]);
        }
        // (reciprocals by Newton steps, ~1 ulp: two IEEE divisions were ~60 dependent instructions
        // of the workgroup's serial path; a zero class sum gives NaN either way)
        const double rbar = norm * fast_rcp(s0k);                                       // eq. 2.5
        double conc = (rbar * E - rbar * rbar * rbar) * fast_rcp(1.0 - rbar * rbar);  // eq. 4.4
        conc = conc < a.cmin ? a.cmin : (conc > a.cmax ? a.cmax : conc);      // NaN stays NaN
        const double lb = bR > 0 ? wave_log_bessel_fixed(btab, nu, conc, lane, bR)
                                 : wave_log_bessel_over_power(nu, conc, lane);
        const double o = -(0.5 * E * kLn2PiB + lb);
        if (lane == 0) {
          sprec[k] = conc;
          soff[k] = o;
          // estimate_mixture_weight with saliency: L1 unit norm, eps 'where' 1e-10
          swgt[k] = (a.weight_mode == 1) ? 1.0 / K : s0k / (tabs == 0.0 ? 1e-10 : tabs);
        }
      }
      __syncthreads();
    }
    if (a.iterations > 0) {
      for (int i = tid; i < K * E; i += NW * kWave) {
        const int k = i / E, e = i - k * E;
        a.out_mean[(size_t)b * K * E + i] = mu[k * 16 + e];
      }
      if (tid < K) {
        a.out_conc[b * K + tid] = sprec[tid];
        a.out_weight[b * K + tid] = swgt[tid];
      }
    }
    if (a.out_aff && wave < nact) {  // final E-step
      double m[K], prec[K], off[K], wgt[K];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        m[k] = mu[k * 16 + (lane & 15)];
        prec[k] = sprec[k];
        off[k] = soff[k];
        wgt[k] = swgt[k];
      }
      for (int c = wave; c < nchunk; c += NW) {
        const int n = c * kWave + lane;
        double v[EP], g[K];
        load_row(tile, c, lane, v);
        posterior(v, rinv[n], m, prec, off, wgt, g);
        if (n < N) {
#pragma unroll
          for (int k = 0; k < K; ++k) a.out_aff[((size_t)b * K + k) * N + n] = g[k];
        }
      }
    }
  }
};

template <int K, int EP, typename TS, int NW>
__global__ void __launch_bounds__(NW* kWave, 2) vmf_bin_em2_kernel(VmfBinArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smraw[];
  VmfBin<K, EP, TS, NW>::run(a, smraw);
}

template <int K, int EP, typename TS, int NW>
int go(const VmfBinArgs& a, int64_t B, size_t lds_limit, hipStream_t s) {
  using Kern = VmfBin<K, EP, TS, NW>;
  const size_t lds = Kern::lds_bytes(a.N);
  if (lds > lds_limit) return PBBSS_ERR_UNSUPPORTED;
  auto kfn = vmf_bin_em2_kernel<K, EP, TS, NW>;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return PBBSS_ERR_HIP;
  hipLaunchKernelGGL(kfn, dim3((unsigned)B), dim3(NW * kWave), lds, s, a);
  return hipGetLastError() == hipSuccess ? PBBSS_OK : PBBSS_ERR_HIP;
}

template <int K, int EP>
int go_k(const VmfBinArgs& a, int y_is_f64, int64_t B, size_t lds_limit, int num_cu, hipStream_t s) {
  if constexpr (K * EP + K > 64) {
    return PBBSS_ERR_UNSUPPORTED;
  } else {
    // Wavefronts per mixture.  More than four chunks of rows: eight waves (two per SIMD) while
    // every mixture gets a compute unit of its own.  With more mixtures than compute units (257
    // bins on 256 CUs) the surplus must be co-hosted or the launch takes a second round for it:
    // four waves then, two workgroups per CU at the kernel's ~175 registers.  (Six waves capped at
    // 168 registers for three per SIMD were measured: the cap costs spills on the serial path,
    // 1.10 ms per 100-iteration fit against 0.64 ms with four waves and 0.45 ms with eight on 256
    // bins -- profiles/r06_c_vmf_bin_variants.txt.)
    const int nchunk = (a.N + kWave - 1) / kWave;
    int nw = (nchunk > 4 && B <= num_cu) ? 8 : 4;
    if (const char* v = getenv("PBBSS_VMF_NW")) nw = atoi(v) >= 8 ? 8 : 4;  // development knob
    if (y_is_f64) {
      if (nw == 8) return go<K, EP, double, 8>(a, B, lds_limit, s);
      return go<K, EP, double, 4>(a, B, lds_limit, s);
    }
    if (nw == 8) return go<K, EP, float, 8>(a, B, lds_limit, s);
    return go<K, EP, float, 4>(a, B, lds_limit, s);
  }
}

template <int K>
int go_e(const VmfBinArgs& a, int y_is_f64, int64_t B, size_t lds_limit, int num_cu, hipStream_t s) {
  const int EP = (a.E + 3) & ~3;
  switch (EP) {
    case 4: return go_k<K, 4>(a, y_is_f64, B, lds_limit, num_cu, s);
    case 8: return go_k<K, 8>(a, y_is_f64, B, lds_limit, num_cu, s);
    case 12: return go_k<K, 12>(a, y_is_f64, B, lds_limit, num_cu, s);
    case 16: return go_k<K, 16>(a, y_is_f64, B, lds_limit, num_cu, s);
    default: return PBBSS_ERR_UNSUPPORTED;
  }
}

}  // namespace

// PBBSS_ERR_UNSUPPORTED: the shape is not served here (E > 16, K outside 2..4, K (E' + 1) > 64
// accumulators, rows beyond the LDS budget) -- the caller takes vmf_bin_em_kernel / the sweeps.
int launch_vmf_bin_em2(const void* y, int y_is_f64, int64_t B, int64_t N, int E, int K,
                       int iterations, const double* gamma, const double* sal,
                       const double* in_mean, const double* in_conc, const double* in_weight,
                       double cmin, double cmax, int weight_mode, double* mean, double* conc,
                       double* weight, double* out_aff, size_t lds_limit, int num_cu,
                       hipStream_t s) {
  static const bool off = [] {  // development knob: PBBSS_VMF_BIN2=0 keeps the round-4 kernel
    const char* v = getenv("PBBSS_VMF_BIN2");
    return v && v[0] == '0';
  }();
  if (off || E < 1 || E > 16 || N < 1 || N > 65536 || B < 1 || B > 2147483647LL)
    return PBBSS_ERR_UNSUPPORTED;
  VmfBinArgs a{y, (int)N, E, iterations, gamma, sal, in_mean, in_conc, in_weight, cmin, cmax,
               weight_mode, mean, conc, weight, out_aff};
  switch (K) {
    case 2: return go_e<2>(a, y_is_f64, B, lds_limit, num_cu, s);
    case 3: return go_e<3>(a, y_is_f64, B, lds_limit, num_cu, s);
    case 4: return go_e<4>(a, y_is_f64, B, lds_limit, num_cu, s);
    default: return PBBSS_ERR_UNSUPPORTED;
  }
}

}  // namespace pbbss
