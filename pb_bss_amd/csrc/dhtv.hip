// DHTV frequency-permutation alignment on the device.
//
// Reference: permutation_alignment.py:295-355 (DHTVPermutationAlignment.
// calculate_mapping, similarity_metric='cos'), :380-407 (score matrix),
// :469-589 (_mapping_from_score_matrix), :54-104 (apply_mapping).
//
// One 1024-thread workgroup per utterance runs the WHOLE alignment plan in one
// launch (the reference's python loops are plan x iteration x frequency).
// Inside one iteration every frequency of the segment is independent given the
// time centroid, so the 16 waves each take frequencies round-robin:
//   centroid  thread = (class, frame) column, sum over the segment's bins
//   scores    wave = frequency, lanes = frames: K x K dot products, butterfly
//   assign    greedy / brute-force optimal on the K x K scores (uniform code)
//   permute   the K feature rows of that frequency and its mapping column
// Features (unit-norm masks, float64) live in a caller-provided scratch copy in
// HBM/L2 (K*F*T*8 bytes per utterance); the normalised centroid in LDS.
#include "dhtv.hpp"
#include "pbbss_dev.hpp"

namespace pbbss {

constexpr int kDhtvThreads = 1024;
constexpr int kDhtvWaves = kDhtvThreads / kWave;
constexpr int kDhtvMaxK = 8;

__device__ __forceinline__ double block_sum(double v, double* red, int tid) {
  // sum over the whole workgroup (red: kDhtvWaves doubles in LDS)
  v = wave_sum(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int w = 0; w < kDhtvWaves; ++w) s += red[w];
  return s;
}

// permutation index -> permutation in itertools.permutations (lexicographic) order
__device__ __forceinline__ void nth_permutation(int n, int K, int* out) {
  int avail[kDhtvMaxK];
  for (int i = 0; i < K; ++i) avail[i] = i;
  int fact = 1;
  for (int i = 2; i < K; ++i) fact *= i;  // (K-1)!
  for (int i = 0; i < K; ++i) {
    int idx = n / fact;
    n %= fact;
    out[i] = avail[idx];
    for (int j = idx; j < K - 1 - i; ++j) avail[j] = avail[j + 1];
    if (K - 1 - i > 0) fact /= (K - 1 - i);
  }
}

// greedy / brute-force optimal assignment on the K x K score matrix (uniform across the wave)
template <int K>
__device__ __forceinline__ void assign_classes(double (&sc)[K][K], int optimal, int (&perm)[K]) {
    if (!optimal) {
      // greedy: repeatedly take the flat (row-major) argmax, first maximum wins (:537-553)
      bool row_used[K], col_used[K];
#pragma unroll
      for (int a = 0; a < K; ++a) {
        row_used[a] = false;
        col_used[a] = false;
        perm[a] = a;
      }
#pragma unroll
      for (int r = 0; r < K; ++r) {
        double best = 0.0;
        int bi = -1, bj = -1;
#pragma unroll
        for (int a = 0; a < K; ++a)
#pragma unroll
          for (int b = 0; b < K; ++b) {
            // masked entries are -inf in the reference: an unmasked one always
            // wins the first comparison; among all-masked (cannot happen) none
            bool ok = !row_used[a] && !col_used[b];
            if (ok && (bi < 0 || sc[a][b] > best)) {
              best = sc[a][b];
              bi = a;
              bj = b;
            }
          }
#pragma unroll
        for (int a = 0; a < K; ++a) {
          if (a == bi) {
            row_used[a] = true;
            perm[a] = bj;
          }
          if (a == bj) col_used[a] = true;
        }
      }
    } else {
      // brute force over itertools.permutations order, strict improvement (:566-586)
      int nperm = 1;
      for (int i = 2; i <= K; ++i) nperm *= i;
      double best = -1.79e308;
#pragma unroll
      for (int a = 0; a < K; ++a) perm[a] = a;
      for (int n = 0; n < nperm; ++n) {
        int p[kDhtvMaxK];
        nth_permutation(n, K, p);
        double v = 0.0;
        for (int a = 0; a < K; ++a) {
          double x = 0.0;
#pragma unroll
          for (int aa = 0; aa < K; ++aa)
#pragma unroll
            for (int bb = 0; bb < K; ++bb)
              if (aa == a && bb == p[a]) x = sc[aa][bb];
          v += x;
        }
        if (v > best) {
          best = v;
#pragma unroll
          for (int a = 0; a < K; ++a) perm[a] = p[a];
        }
      }
    }
}

template <int K>
__global__ void __launch_bounds__(kDhtvThreads)
    dhtv_kernel(const double* __restrict__ mask, double* __restrict__ feat_all,
                int32_t* __restrict__ mapping_all, const int32_t* __restrict__ plan, int P, int F,
                int T, int optimal, int metric, int32_t* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* cent = reinterpret_cast<double*>(smem);  // [K][T]
  double* red = cent + (size_t)K * T;              // [kDhtvWaves]
  int* flag = reinterpret_cast<int*>(red + kDhtvWaves);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t u = blockIdx.x;
  const double* m = mask + u * (int64_t)K * F * T;
  double* feat = feat_all + u * (int64_t)K * F * T;
  int32_t* mapping = mapping_all + u * (int64_t)K * F;

  // features = mask / max(||mask||, tiny) per (class, bin) row  (:310, :358-377)
  int nonfinite = 0;
  for (int row = wave; row < K * F; row += kDhtvWaves) {
    const double* src = m + (int64_t)row * T;
    double ss = 0.0;
    for (int t = lane; t < T; t += kWave) {
      double v = src[t];
      ss += v * v;
    }
    ss = wave_sum(ss);
    if (!isfinite(ss)) nonfinite = 1;
    // 'cos': unit-norm rows; other metrics: features = mask.copy() (:309-312)
    double inv = (metric == PBBSS_PA_COS) ? 1.0 / fmax(sqrt(ss), kTiny) : 1.0;
    double* dst = feat + (int64_t)row * T;
    for (int t = lane; t < T; t += kWave) dst[t] = src[t] * inv;
  }
  for (int i = tid; i < K * F; i += kDhtvThreads) mapping[i] = i / F;
  if (tid == 0) *flag = 0;
  __syncthreads();

  for (int seg = 0; seg < P; ++seg) {
    const int iterations = plan[3 * seg], start = plan[3 * seg + 1], end = plan[3 * seg + 2];
    const double inv_n = 1.0 / (double)(end - start);
    for (int it = 0; it < iterations; ++it) {
      // time centroid = mean over the segment's bins, then unit norm per class (:334-340)
      // (eight independent partial sums: the loads of a column are L2 round trips, a
      // single running sum would serialise ~100 of them per thread)
      for (int col = tid; col < K * T; col += kDhtvThreads) {
        const int k = col / T, t = col - k * T;
        const double* p = feat + ((int64_t)k * F + start) * T + t;
        double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int f = start;
        for (; f + 8 <= end; f += 8, p += 8 * (int64_t)T) {
#pragma unroll
          for (int u = 0; u < 8; ++u) s[u] += p[u * (int64_t)T];
        }
        for (; f < end; ++f, p += T) s[0] += *p;
        cent[col] = (((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]))) * inv_n;
      }
      __syncthreads();
      for (int k = 0; k < K && metric == PBBSS_PA_COS; ++k) {  // :337-341
        double ss = 0.0;
        for (int t = tid; t < T; t += kDhtvThreads) {
          double v = cent[k * T + t];
          ss += v * v;
        }
        ss = block_sum(ss, red, tid);
        double inv = 1.0 / fmax(sqrt(ss), kTiny);
        for (int t = tid; t < T; t += kDhtvThreads) cent[k * T + t] *= inv;
      }
      __syncthreads();
      // every frequency of the segment against the centroid
      int changed = 0;
      for (int f = start + wave; f < end; f += kDhtvWaves) {
        double sc[K][K];  // [reference class][mask class]  ('K...T,k...T->...kK')
#pragma unroll
        for (int a = 0; a < K; ++a)
#pragma unroll
          for (int b = 0; b < K; ++b) sc[a][b] = 0.0;
        // four frame-steps per trip: all their global loads are issued before the first
        // FMA so the L2 round trips overlap (a 1-step loop serialises 8 of them per bin)
        for (int t0 = lane; t0 < T; t0 += 4 * kWave) {
          double fv[4][K], cv[4][K];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int t = t0 + u * kWave;
            const bool ok = t < T;
            const int tc = ok ? t : lane;
#pragma unroll
            for (int k = 0; k < K; ++k) {
              double v = feat[((int64_t)k * F + f) * T + tc];
              fv[u][k] = ok ? v : 0.0;
              cv[u][k] = ok ? cent[k * T + tc] : 0.0;
            }
          }
          if (metric == PBBSS_PA_EUCLIDEAN) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
              for (int a = 0; a < K; ++a)
#pragma unroll
                for (int b = 0; b < K; ++b) {
                  double d = fv[u][b] - cv[u][a];
                  sc[a][b] = fma(d, d, sc[a][b]);
                }
          } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
              for (int a = 0; a < K; ++a)
#pragma unroll
                for (int b = 0; b < K; ++b) sc[a][b] = fma(cv[u][a], fv[u][b], sc[a][b]);
          }
        }
        bool finite = true;
#pragma unroll
        for (int a = 0; a < K; ++a)
#pragma unroll
          for (int b = 0; b < K; ++b) {
            sc[a][b] = wave_sum(sc[a][b]);
            if (metric == PBBSS_PA_EUCLIDEAN) sc[a][b] = -sqrt(sc[a][b]);  // :412-416
            finite = finite && isfinite(sc[a][b]);
          }
        if (!finite) nonfinite = 1;  // reference: ValueError('score matrix is infeasible')
        int perm[K];
        assign_classes<K>(sc, optimal, perm);
        bool ident = true;
#pragma unroll
        for (int a = 0; a < K; ++a) ident = ident && (perm[a] == a);
        if (!ident) {
          changed = 1;
          // features[:, f, :] = features[perm, f, :]; mapping[:, f] = mapping[perm, f] (:348-350)
          for (int t = lane; t < T; t += kWave) {
            double v[K];
#pragma unroll
            for (int k = 0; k < K; ++k) v[k] = feat[((int64_t)k * F + f) * T + t];
#pragma unroll
            for (int a = 0; a < K; ++a) {
              double x = v[0];
#pragma unroll
              for (int k = 1; k < K; ++k) x = (perm[a] == k) ? v[k] : x;
              feat[((int64_t)a * F + f) * T + t] = x;
            }
          }
          if (lane == 0) {
            int mv[K];
#pragma unroll
            for (int k = 0; k < K; ++k) mv[k] = mapping[k * F + f];
#pragma unroll
            for (int a = 0; a < K; ++a) {
              int x = mv[0];
#pragma unroll
              for (int k = 1; k < K; ++k) x = (perm[a] == k) ? mv[k] : x;
              mapping[a * F + f] = x;
            }
          }
        }
      }
      if (changed && lane == 0) atomicOr(flag, 1);
      __syncthreads();
      const int any = *flag;
      __syncthreads();
      if (tid == 0) *flag = 0;
      __syncthreads();
      if (!any) break;  // nothing_changed (:352-353)
    }
  }
  if (nonfinite && lane == 0) atomicOr(status + u, (int32_t)PBBSS_ST_NONFINITE);
}

// ---------------------------------------------------------------------------------------
// Team variant for FEW utterances: the single-workgroup kernel above is bound by ONE compute
// unit's L2 bandwidth and latency (2.8 ms for F=513, T=500, K=3 -- longer than the 100 EM
// iterations that produced the masks).  Here G workgroups share one utterance:
//   A  partial time centroids: workgroup (column block, bin chunk) sums its bins of the
//      segment for 1024 (class, frame) columns -> partial buffer in L2
//   -- team barrier --
//      every workgroup adds the chunk partials in a fixed order into its LDS centroid and
//      normalises it (redundantly: 12 KB, cheaper than another exchange)
//   B  one wavefront per frequency bin of the segment: scores, assignment, row permutation
//   -- team barrier (also carries the "something changed" flag) --
// Cross-workgroup data (features, mapping, partials, flags) move with write-through (sc1)
// stores and L1-bypassing (sc1) loads; the barrier is a monotonic agent-scope counter with a
// bounded spin (cdna_hip_programming.md guideline 16, form R1) -- same protocol as the
// split-bin EM groups.  Needs all G workgroups of an utterance co-resident: G <= 32.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double ld_sc1(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int ld_sc1(const int32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(int32_t* p, int v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr unsigned kTeamSpinLimit = 20000000u;

// ctrl: [0] barrier counter, [1] error word, [2], [3] "changed" flags by iteration parity
// spin_limit: polls before the barrier gives up (a kernel argument: the handle's
// pbbss_set_spin_limit value -- it was a device global shared by every handle until round 6);
// 0 = kTeamSpinLimit
__device__ __forceinline__ void team_barrier(unsigned* ctrl, unsigned& target, int G, int tid,
                                             unsigned spin_limit) {
  target += (unsigned)G;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's sc1 stores have reached L2
  __syncthreads();
  if (tid == 0) {
    __hip_atomic_fetch_add(ctrl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (__hip_atomic_load(ctrl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      // a timed-out wait is sticky: every later barrier of the team falls through at once
      // (the result is reported as failed), so a lost workgroup can never hang the device
      if ((spins & 1023u) == 1023u &&
          __hip_atomic_load(ctrl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
        break;
      if (++spins >= (spin_limit ? spin_limit : kTeamSpinLimit)) {
        __hip_atomic_store(ctrl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
  __syncthreads();
}

template <int K>
__global__ void __launch_bounds__(kDhtvThreads)
    dhtv_team_kernel(const double* __restrict__ mask, double* feat_all, int32_t* mapping_all,
                     const int32_t* __restrict__ plan, int P, int F, int T, int optimal,
                     int metric, int32_t* status, int G, double* part_all, unsigned* ctrl_all,
                     unsigned spin_limit) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* cent = reinterpret_cast<double*>(smem);  // [K][T]
  double* red = cent + (size_t)K * T;              // [kDhtvWaves]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t u = blockIdx.x / G;
  const int g = blockIdx.x % G;
  const double* m = mask + u * (int64_t)K * F * T;
  double* feat = feat_all + u * (int64_t)K * F * T;
  int32_t* mapping = mapping_all + u * (int64_t)K * F;
  const int KT = K * T;
  const int ncb = (KT + kDhtvThreads - 1) / kDhtvThreads;  // column blocks
  const int NCH = G / ncb > 0 ? G / ncb : 1;               // bin chunks (G >= ncb by launch)
  double* part = part_all + u * (int64_t)NCH * KT;
  unsigned* ctrl = ctrl_all + u * 4;
  unsigned target = 0;

  int nonfinite = 0;
  for (int row = g * kDhtvWaves + wave; row < K * F; row += G * kDhtvWaves) {
    const double* src = m + (int64_t)row * T;
    double ss = 0.0;
    for (int t = lane; t < T; t += kWave) {
      double v = src[t];
      ss += v * v;
    }
    ss = wave_sum(ss);
    if (!isfinite(ss)) nonfinite = 1;
    // 'cos': unit-norm rows; other metrics: features = mask.copy() (:309-312)
    double inv = (metric == PBBSS_PA_COS) ? 1.0 / fmax(sqrt(ss), kTiny) : 1.0;
    double* dst = feat + (int64_t)row * T;
    for (int t = lane; t < T; t += kWave) st_sc1(dst + t, src[t] * inv);
  }
  if (g == 0)
    for (int i = tid; i < K * F; i += kDhtvThreads) st_sc1(mapping + i, i / F);
  team_barrier(ctrl, target, G, tid, spin_limit);

  int itn = 0;  // global iteration counter (parity of the changed flag)
  for (int seg = 0; seg < P; ++seg) {
    const int iterations = plan[3 * seg], start = plan[3 * seg + 1], end = plan[3 * seg + 2];
    const double inv_n = 1.0 / (double)(end - start);
    const int L = (end - start + NCH - 1) / NCH;
    for (int it = 0; it < iterations; ++it, ++itn) {
      // ---- A: partial centroid of (column block cb, bin chunk c)
      {
        const int cb = g % ncb, c = g / ncb;
        const int col = cb * kDhtvThreads + tid;
        if (c < NCH && col < KT) {
          const int k = col / T, t = col - k * T;
          const int f0 = start + c * L, f1 = (f0 + L < end) ? f0 + L : end;
          const double* p = feat + ((int64_t)k * F + f0) * T + t;
          double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
          int f = f0;
          for (; f + 8 <= f1; f += 8, p += 8 * (int64_t)T) {
            // all eight (atomic, hence program-ordered) loads first, then the adds: written as
            // load-add pairs every add waits for its own L2 round trip
            double a[8];
#pragma unroll
            for (int x = 0; x < 8; ++x) a[x] = ld_sc1(p + x * (int64_t)T);
#pragma unroll
            for (int x = 0; x < 8; ++x) s[x] += a[x];
          }
          for (; f < f1; ++f, p += T) s[0] += ld_sc1(p);
          st_sc1(part + (int64_t)c * KT + col,
                 ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7])));
        }
      }
      team_barrier(ctrl, target, G, tid, spin_limit);
      if (g == 0 && tid == 0) st_sc1(reinterpret_cast<int32_t*>(ctrl + 2 + ((itn + 1) & 1)), 0);
      // ---- centroid = ordered sum of the chunk partials, mean, unit norm per class
      for (int col = tid; col < KT; col += kDhtvThreads) {
        double sacc = 0.0;
        int c = 0;
        for (; c + 4 <= NCH; c += 4) {  // same summation order, four round trips overlapped
          double a[4];
#pragma unroll
          for (int x = 0; x < 4; ++x) a[x] = ld_sc1(part + (int64_t)(c + x) * KT + col);
#pragma unroll
          for (int x = 0; x < 4; ++x) sacc += a[x];
        }
        for (; c < NCH; ++c) sacc += ld_sc1(part + (int64_t)c * KT + col);
        cent[col] = sacc * inv_n;
      }
      __syncthreads();
      for (int k = 0; k < K && metric == PBBSS_PA_COS; ++k) {  // :337-341
        double ss = 0.0;
        for (int t = tid; t < T; t += kDhtvThreads) {
          double v = cent[k * T + t];
          ss += v * v;
        }
        ss = block_sum(ss, red, tid);
        double inv = 1.0 / fmax(sqrt(ss), kTiny);
        for (int t = tid; t < T; t += kDhtvThreads) cent[k * T + t] *= inv;
      }
      __syncthreads();
      // ---- B: one wavefront per bin
      int changed = 0;
      for (int f = start + g * kDhtvWaves + wave; f < end; f += G * kDhtvWaves) {
        double sc[K][K];
#pragma unroll
        for (int a = 0; a < K; ++a)
#pragma unroll
          for (int b = 0; b < K; ++b) sc[a][b] = 0.0;
        for (int t0 = lane; t0 < T; t0 += 4 * kWave) {
          double fv[4][K], cv[4][K];
#pragma unroll
          for (int x = 0; x < 4; ++x) {  // raw loads first (clamped frames), masks afterwards
            const int t = t0 + x * kWave;
            const int tc = (t < T) ? t : lane;
#pragma unroll
            for (int k = 0; k < K; ++k) fv[x][k] = ld_sc1(feat + ((int64_t)k * F + f) * T + tc);
          }
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            const int t = t0 + x * kWave;
            const bool ok = t < T;
            const int tc = ok ? t : lane;
#pragma unroll
            for (int k = 0; k < K; ++k) {
              fv[x][k] = ok ? fv[x][k] : 0.0;
              cv[x][k] = ok ? cent[k * T + tc] : 0.0;
            }
          }
          if (metric == PBBSS_PA_EUCLIDEAN) {
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
              for (int a = 0; a < K; ++a)
#pragma unroll
                for (int b = 0; b < K; ++b) {
                  double d = fv[x][b] - cv[x][a];
                  sc[a][b] = fma(d, d, sc[a][b]);
                }
          } else {
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
              for (int a = 0; a < K; ++a)
#pragma unroll
                for (int b = 0; b < K; ++b) sc[a][b] = fma(cv[x][a], fv[x][b], sc[a][b]);
          }
        }
        bool finite = true;
#pragma unroll
        for (int a = 0; a < K; ++a)
#pragma unroll
          for (int b = 0; b < K; ++b) {
            sc[a][b] = wave_sum(sc[a][b]);
            if (metric == PBBSS_PA_EUCLIDEAN) sc[a][b] = -sqrt(sc[a][b]);  // :412-416
            finite = finite && isfinite(sc[a][b]);
          }
        if (!finite) nonfinite = 1;
        int perm[K];
        assign_classes<K>(sc, optimal, perm);
        bool ident = true;
#pragma unroll
        for (int a = 0; a < K; ++a) ident = ident && (perm[a] == a);
        if (!ident) {
          changed = 1;
          for (int t = lane; t < T; t += kWave) {
            double v[K];
#pragma unroll
            for (int k = 0; k < K; ++k) v[k] = ld_sc1(feat + ((int64_t)k * F + f) * T + t);
#pragma unroll
            for (int a = 0; a < K; ++a) {
              double x = v[0];
#pragma unroll
              for (int k = 1; k < K; ++k) x = (perm[a] == k) ? v[k] : x;
              st_sc1(feat + ((int64_t)a * F + f) * T + t, x);
            }
          }
          if (lane == 0) {
            int mv[K];
#pragma unroll
            for (int k = 0; k < K; ++k) mv[k] = ld_sc1(mapping + k * F + f);
#pragma unroll
            for (int a = 0; a < K; ++a) {
              int x = mv[0];
#pragma unroll
              for (int k = 1; k < K; ++k) x = (perm[a] == k) ? mv[k] : x;
              st_sc1(mapping + a * F + f, x);
            }
          }
        }
      }
      if (changed && lane == 0)
        __hip_atomic_fetch_or(ctrl + 2 + (itn & 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      team_barrier(ctrl, target, G, tid, spin_limit);
      const unsigned any =
          __hip_atomic_load(ctrl + 2 + (itn & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (!any) {
        ++itn;
        break;  // nothing_changed (:352-353)
      }
    }
  }
  const unsigned xerr = __hip_atomic_load(ctrl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if ((nonfinite || xerr) && lane == 0)
    atomicOr(status + u, (int32_t)(nonfinite ? PBBSS_ST_NONFINITE : 0) |
                             (int32_t)(xerr ? PBBSS_ST_EIG_NOCONV : 0));
}

// ---------------------------------------------------------------------------------------
// Frame-slice variant: the features never move.
//
// The two kernels above permute feature rows in memory, so every iteration pays L2 round
// trips for data other workgroups wrote (team kernel: two team barriers and four dependent
// round trips, ~20 us per plan iteration).  Here the G workgroups of an utterance split the
// FRAME axis instead of the bins: workgroup g owns frames [g*TS, (g+1)*TS), TS = 16*NFR, of
// EVERY bin, reads them from the (read-only) mask once per segment and keeps them in
// registers -- 16-lane row = bin of the segment (32 per pass), lane = NFR frames.  Then
//   * the time centroid of its frames is LOCAL (sum over the segment's bins, no exchange);
//   * the K x K scores and the centroid norms are sums over frames: each workgroup publishes
//     its partial (bins x K*K + K doubles), ONE team barrier, and every workgroup adds the G
//     partials in the same order and solves every assignment itself -- all workgroups reach
//     identical permutations, the "anything changed" decision needs no flag;
//   * a permutation is a register shuffle in the owning wave plus a 4-bit-per-class word in
//     LDS (the composite mapping).
// One exchange hop per plan iteration instead of four.  'cos': the centroid norm is applied
// to the summed score (score * 1/max(||c||, tiny)) instead of to the centroid before the dot
// product -- same value up to one rounding.
// The row scales come from dhtv_rowscale_kernel (whole chip, one wavefront per row) and the
// aligned features are written by dhtv_features_kernel afterwards: inside this kernel both were
// 25 + 31 us on G compute units.  The exchange area is the head of the utterance's feature
// scratch (dead until dhtv_features_kernel): 2 parities x G x (K + F*K*K) doubles.
//
// Measured and rejected (profiles/r03_h_dhtv.txt): the exchange without a barrier (every double
// as two 64-bit words carrying a sequence number, readers polling the data itself) -- the
// polling traffic costs more than the barrier saves; 8-lane rows x 8 consecutive frames (half
// the butterflies, but 64-byte per-lane reads and an eight-row centroid reduction); confining a
// team to one XCD (blockIdx % 8): agent-scope accesses do not get faster, workgroup-scope ones
// (sc0) are served by the CU's L1 and never see the peers' stores.
// ---------------------------------------------------------------------------------------
constexpr int kSliceThreads = 512;  // 8 waves: 256 VGPRs per lane keep the window in registers
constexpr int kSliceWaves = kSliceThreads / kWave;
constexpr int kSliceBins = kSliceThreads / 16;  // bins per pass: one per 16-lane row

template <int K>
__device__ __forceinline__ int pack_identity() {
  int m = 0;
#pragma unroll
  for (int a = 0; a < K; ++a) m |= a << (4 * a);
  return m;
}

// Lane layout: 16-lane row = one bin, the row's lanes = 16 consecutive frames, NFR frames per
// lane (t = t0 + l + 16 j): the frame sums of the scores stay inside a row (DPP only, no
// permlane swaps), 32 bins per pass.
template <int K, int NFR>
struct SliceCfg {
  static constexpr int KK = K * K;
  static constexpr int NS = ((KK + 3) / 4) * 4;  // scores per row reduce-scatter
  static constexpr int QS = NS / 4;
  static constexpr int NC = K * NFR;             // centroid values per lane (NFR % 4 == 0)
  static constexpr int QC = NC / 4;
  static constexpr int kRegDoubles = 48;         // register-resident window budget
  static constexpr int MAXP = (kRegDoubles / NC) < 1 ? 1 : ((kRegDoubles / NC) > 4 ? 4 : (kRegDoubles / NC));
  static constexpr int TS = 16 * NFR;
};

// sum over the 16 lanes of a row, halving: on return v[0 .. N/4) hold the totals of the original
// indices (N/4) * (b2 + 2*b3) + m (b_i = lane bit i; the 4 lanes of a quad hold the same values)
template <int N>
__device__ __forceinline__ void row_reduce_scatter(double (&v)[N]) {
  static_assert(N % 4 == 0, "pad to a multiple of 4");
  {
    constexpr int H = N / 2;
#pragma unroll
    for (int n = 0; n < H; ++n) {
      double lo = v[n], hi = v[n + H];
      double recv = dpp_f64<kDppRowRor8, 0x3>(lo, lo);
      recv = dpp_f64<kDppRowRor8, 0xC>(recv, hi);
      double keep = dpp_f64<kDppQuadIdent, 0xC>(lo, hi);
      v[n] = keep + recv;
    }
  }
  constexpr int Q = N / 4;
#pragma unroll
  for (int n = 0; n < Q; ++n) {
    double lo = v[n], hi = v[n + Q];
    double recv = dpp_f64<kDppRowRor12, 0x5>(lo, lo);
    recv = dpp_f64<kDppRowRor4, 0xA>(recv, hi);
    double keep = dpp_f64<kDppQuadIdent, 0xA>(lo, hi);
    v[n] = keep + recv;
  }
#pragma unroll
  for (int n = 0; n < Q; ++n) v[n] += dpp_f64<kDppQuadXor2, 0xF>(v[n], v[n]);
#pragma unroll
  for (int n = 0; n < Q; ++n) v[n] += dpp_f64<kDppQuadXor1, 0xF>(v[n], v[n]);
}
// sum over the four rows of a wave (lane bits 5, 4), halving: v[0 .. N/4) = totals of the original
// indices (N/4) * (b4 + 2*b5) + m
template <int N>
__device__ __forceinline__ void rows_reduce_scatter(double (&v)[N]) {
  static_assert(N % 4 == 0, "pad to a multiple of 4");
  {
    constexpr int H = N / 2;
#pragma unroll
    for (int n = 0; n < H; ++n) {
      double lo = v[n], hi = v[n + H];
      swap32_f64(lo, hi);
      v[n] = lo + hi;
    }
  }
  constexpr int Q = N / 4;
#pragma unroll
  for (int n = 0; n < Q; ++n) {
    double lo = v[n], hi = v[n + Q];
    swap16_f64(lo, hi);
    v[n] = lo + hi;
  }
}
__device__ __forceinline__ double row_sum(double v) {
  v += dpp_f64<kDppQuadXor1, 0xF>(v, v);
  v += dpp_f64<kDppQuadXor2, 0xF>(v, v);
  v += dpp_f64<kDppRowHalfMirror, 0xF>(v, v);
  v += dpp_f64<kDppRowMirror, 0xF>(v, v);
  return v;
}

__host__ __device__ inline size_t slice_entries(int K, int F) {
  return (((size_t)K + (size_t)F * K * K) + 7) & ~(size_t)7;
}
// LDS: centroid [K][TS], row scales [K][F], union{wave sums [waves][K][TS], summed entries},
// composite mapping [F], last permutation [F]
inline size_t slice_lds_bytes(int K, int NFR, int F) {
  const size_t TS = (size_t)16 * NFR;
  const size_t uni = (size_t)kSliceWaves * K * TS > slice_entries(K, F) ? (size_t)kSliceWaves * K * TS
                                                                        : slice_entries(K, F);
  return ((size_t)K * TS + (size_t)K * F + uni) * sizeof(double) + 2 * (size_t)F * sizeof(int) + 16;
}

// 1 / max(||mask[u,k,f,:]||, tiny) per row ('cos'; 1 otherwise) -- :310, :358-377 -- one wavefront
// per row, the whole chip instead of the G workgroups of the slice kernel
__global__ void __launch_bounds__(256)
    dhtv_rowscale_kernel(const double* __restrict__ mask, int64_t rows, int T, int KF, int cos,
                         double* __restrict__ scale, int32_t* __restrict__ status,
                         unsigned* __restrict__ ctrl, int nctrl) {
  // the slice kernel's control words start at zero (saves a memset launch in front)
  if (blockIdx.x == 0)
    for (int i = threadIdx.x; i < nctrl; i += blockDim.x) ctrl[i] = 0u;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= rows) return;
  const double* src = mask + r * T;
  double s[4] = {0, 0, 0, 0};
  int t = lane;
  for (; t + 3 * kWave < T; t += 4 * kWave) {
    double a[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) a[x] = src[t + x * kWave];
#pragma unroll
    for (int x = 0; x < 4; ++x) s[x] = fma(a[x], a[x], s[x]);
  }
  for (; t < T; t += kWave) s[0] = fma(src[t], src[t], s[0]);
  const double ss = wave_sum((s[0] + s[1]) + (s[2] + s[3]));
  if (lane == 0) {
    if (!isfinite(ss)) atomicOr(status + r / KF, (int32_t)PBBSS_ST_NONFINITE);
    scale[r] = cos ? 1.0 / fmax(sqrt(ss), kTiny) : 1.0;
  }
}

// aligned features out[u,a,f,:] = mask[u,mapping[u,a,f],f,:] * scale (:348-350 applied at once)
__global__ void __launch_bounds__(256)
    dhtv_features_kernel(const double* __restrict__ mask, const int32_t* __restrict__ mapping,
                         const double* __restrict__ scale, int K, int F, int T,
                         double* __restrict__ out) {
  const int64_t row = blockIdx.x;  // (u, a, f)
  const int64_t u = row / ((int64_t)K * F);
  const int f = (int)(row % F);
  const int64_t srow = (u * K + mapping[row]) * F + f;
  const double sc = scale[srow];
  const double* src = mask + srow * (int64_t)T;
  double* dst = out + row * (int64_t)T;
  for (int t = threadIdx.x; t < T; t += blockDim.x) dst[t] = src[t] * sc;
}

//
// PROBE instantiation (pbbss_set_dhtv_probe): masks that are ALREADY aligned -- every call of an
// inline aligner after the EM has settled (cacgmm.py:260-267) -- still walk the plan segment by
// segment, one exchange hop each (20 hops = 0.33 ms at F = 513).  But as long as nothing changes
// the features never move, so the first iteration of EVERY segment can be evaluated on the
// input as it stands, all segments at once on P teams of G workgroups: if no bin of any segment
// asks for a permutation, the sequential walk would have found the same ("nothing_changed" in
// each segment, :352-353) and the mapping is the identity.  Per segment: the probe raises
// flag[u][segment] when a bin asks for a permutation (or when it cannot tell: non-finite scores,
// a timed-out wait, no room for its exchange area); the plan kernel proper skips every segment
// whose flag is down as long as none of its bins has been permuted by an earlier segment (the
// composite mapping of all its bins is still the identity: the segment then sees exactly the
// probe's input), and leaves at once when no flag is up.  A handful of flipped bins -- the usual
// state between two EM iterations -- costs the segments around them only.
// Same arithmetic as the plan kernel (same code), so both take identical decisions.
template <int K, int NFR, bool PROBE>
__global__ void __launch_bounds__(kSliceThreads)
    dhtv_slice_kernel(const double* __restrict__ mask, const double* __restrict__ scale_all,
                      double* feat_all, int32_t* mapping_all, const int32_t* __restrict__ plan,
                      int P, int F, int T, int optimal, int metric, int32_t* status, int G,
                      unsigned* ctrl_all, unsigned* probe_flag, unsigned spin_limit) {
  using C = SliceCfg<K, NFR>;
  constexpr int KK = C::KK, NS = C::NS, QS = C::QS, NC = C::NC, QC = C::QC, MAXP = C::MAXP,
                TS = C::TS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l = lane & 15, q = lane >> 4;  // frame lane inside the row, row (= bin) inside the wave
  const int rowid = wave * 4 + q;          // bin slot of this row inside a pass
  const int64_t team = blockIdx.x / G;  // PROBE: (utterance, segment); else the utterance
  const int64_t u = PROBE ? team / P : team;
  const int only_seg = PROBE ? (int)(team % P) : -1;
  const int g = blockIdx.x % G;
  const int t0 = g * TS;
  const double* m = mask + u * (int64_t)K * F * T;
  double* feat = feat_all + u * (int64_t)K * F * T;
  int32_t* mapping = mapping_all + u * (int64_t)K * F;
  unsigned* ctrl = ctrl_all + team * 4;
  // plan kernel: bit s = the probe saw a change in segment s (or could not tell); all ones
  // without a probe.  A segment whose bit is down and whose bins still carry the identity
  // mapping has exactly the probe's input: its first iteration would change nothing -> skipped.
  unsigned long long seg_changed = ~0ull;
  if constexpr (!PROBE) {
    if (probe_flag) {
      seg_changed = 0ull;
      for (int sg = 0; sg < P; ++sg)  // P <= 64 (launch_slice)
        if (__hip_atomic_load(probe_flag + u * P + sg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
          seg_changed |= 1ull << sg;
      if (seg_changed == 0ull) {
        // nothing to permute in any segment: identity (non-finite rows were already reported
        // by dhtv_rowscale_kernel)
        if (g == 0)
          for (int i = tid; i < K * F; i += kSliceThreads) mapping[i] = i / F;
        return;
      }
    }
  }
  unsigned* my_flag = PROBE ? probe_flag + team : nullptr;
  size_t NE = slice_entries(K, F);
  const size_t uni_n = (size_t)kSliceWaves * K * TS > NE ? (size_t)kSliceWaves * K * TS : NE;
  double* cent = reinterpret_cast<double*>(smem);  // [K][TS]
  double* inv = cent + (size_t)K * TS;             // [K][F]
  double* uni = inv + (size_t)K * F;               // wave sums | summed exchange entries
  int* mapw = reinterpret_cast<int*>(uni + uni_n);  // [F] composite mapping, 4 bits per class
  int* lastp = mapw + F;                            // [F] permutation of the last assignment (0 = none)
  double* xch = feat;                               // [2][G][NE]
  if constexpr (PROBE) {
    // one exchange area [G][NE(segment)] per segment, packed in plan order inside the
    // utterance's scratch
    size_t off = 0, mine_ne = 8;
    for (int sg = 0; sg <= only_seg; ++sg) {
      const int nbs = min(plan[3 * sg + 2], F) - max(plan[3 * sg + 1], 0);
      mine_ne = slice_entries(K, nbs > 0 ? nbs : 0);
      if (sg < only_seg) off += (size_t)G * mine_ne;
    }
    if (off + (size_t)G * mine_ne > (size_t)K * F * T) {  // no room: cannot tell
      if (tid == 0) __hip_atomic_store(my_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;  // the whole team takes this branch
    }
    xch = feat + off;
    NE = mine_ne;
  }
  unsigned target = 0;
  int hop = 0;
  int nonfinite = 0;
  const bool cos = metric == PBBSS_PA_COS;

  // frames of this lane (clamped address + validity)
  int tc[NFR];
  bool tok[NFR];
#pragma unroll
  for (int j = 0; j < NFR; ++j) {
    const int t = t0 + l + 16 * j;
    tok[j] = t < T;
    tc[j] = tok[j] ? t : T - 1;
  }

  // sum over the G partials of entries [0, n): every workgroup, same order
  auto gather = [&](int par, int n) {
    const double* base = xch + (size_t)par * G * NE;
    for (int e = tid; e < n; e += kSliceThreads) {
      double s = 0.0;
      int gg = 0;
      for (; gg + 8 <= G; gg += 8) {
        double a[8];
#pragma unroll
        for (int x = 0; x < 8; ++x) a[x] = ld_sc1(base + (size_t)(gg + x) * NE + e);
#pragma unroll
        for (int x = 0; x < 8; ++x) s += a[x];
      }
      for (; gg < G; ++gg) s += ld_sc1(base + (size_t)gg * NE + e);
      uni[e] = s;
    }
  };

  // row scales (1 / max(||row||, tiny) for 'cos') from dhtv_rowscale_kernel
  {
    const double* scale = scale_all + u * (int64_t)K * F;
    for (int r = tid; r < K * F; r += kSliceThreads) inv[r] = scale[r];
    for (int f = tid; f < F; f += kSliceThreads) {
      mapw[f] = pack_identity<K>();
      lastp[f] = 0;
    }
    __syncthreads();
  }

  // features of bin f (composite mapping applied), this lane's frames
  auto load_bin = [&](int f, bool valid, double (&out)[K][NFR]) {
    const int fc = valid ? f : 0;
    const int mw = mapw[fc];
#pragma unroll
    for (int a = 0; a < K; ++a) {
      const int ka = (mw >> (4 * a)) & 15;
      const double* row = m + ((int64_t)ka * F + fc) * T;
#pragma unroll
      for (int j = 0; j < NFR; ++j) out[a][j] = row[tc[j]];
    }
#pragma unroll
    for (int a = 0; a < K; ++a) {
      const int ka = (mw >> (4 * a)) & 15;
      const double sc = inv[ka * F + fc];
#pragma unroll
      for (int j = 0; j < NFR; ++j) out[a][j] = (valid && tok[j]) ? out[a][j] * sc : 0.0;
    }
  };

  for (int seg = PROBE ? only_seg : 0; seg < (PROBE ? only_seg + 1 : P); ++seg) {
    // segments are clipped to the bins that exist (the Python layer asserts it; a raw C caller may not)
    const int iterations = PROBE ? min(plan[3 * seg], 1) : plan[3 * seg],
              start = max(plan[3 * seg + 1], 0), end = min(plan[3 * seg + 2], F);
    const int nb = end - start;
    if (nb <= 0) continue;
    if constexpr (!PROBE) {
      if (seg < 64 && !((seg_changed >> seg) & 1ull)) {
        int clean = 1;
        for (int f = start + tid; f < end; f += kSliceThreads)
          clean &= mapw[f] == pack_identity<K>();
        if (__syncthreads_and(clean)) continue;  // every workgroup of the team decides alike
      }
    }
    const double inv_n = 1.0 / (double)nb;
    const int npass = (nb + kSliceBins - 1) / kSliceBins;
    // register-resident part of the window
    double fv[MAXP][K][NFR];
#pragma unroll
    for (int p = 0; p < MAXP; ++p) {
      const int f = start + p * kSliceBins + rowid;
      load_bin(f, f < end, fv[p]);
    }
    for (int it = 0; it < iterations; ++it) {
      double* mine = xch + ((size_t)(hop & 1) * G + g) * NE;
      // ---- time centroid of my frames: sum over the segment's bins (:334-336), local
      {
        double cs[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) cs[c] = 0.0;
#pragma unroll
        for (int p = 0; p < MAXP; ++p)
#pragma unroll
          for (int a = 0; a < K; ++a)
#pragma unroll
            for (int j = 0; j < NFR; ++j) cs[a * NFR + j] += fv[p][a][j];
        for (int p = MAXP; p < npass; ++p) {
          const int f = start + p * kSliceBins + rowid;
          double x[K][NFR];
          load_bin(f, f < end, x);
#pragma unroll
          for (int a = 0; a < K; ++a)
#pragma unroll
            for (int j = 0; j < NFR; ++j) cs[a * NFR + j] += x[a][j];
        }
        rows_reduce_scatter<NC>(cs);  // the wave's four bins; row q keeps values QC*q + i
#pragma unroll
        for (int i = 0; i < QC; ++i) {
          const int c = QC * q + i, a = c / NFR, j = c - a * NFR;
          uni[((size_t)wave * K + a) * TS + l + 16 * j] = cs[i];
        }
      }
      __syncthreads();
      for (int c = tid; c < K * TS; c += kSliceThreads) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < kSliceWaves; ++w) s += uni[(size_t)w * K * TS + c];
        cent[c] = s * inv_n;
      }
      __syncthreads();
      if (wave < K) {  // partial squared norm of class `wave`'s centroid (:337-341)
        double ss = 0.0;
        for (int t = lane; t < TS; t += kWave) {
          const double w = cent[wave * TS + t];
          ss += w * w;
        }
        ss = wave_sum(ss);
        if (lane == 0) st_sc1(mine + wave, ss);
      }
      // ---- partial scores of every bin of the segment over my frames
      auto score_bin = [&](int f, const double (&x)[K][NFR]) {
        double v[NS];
#pragma unroll
        for (int e = 0; e < NS; ++e) v[e] = 0.0;
#pragma unroll
        for (int a = 0; a < K; ++a)
#pragma unroll
          for (int b = 0; b < K; ++b)
#pragma unroll
            for (int j = 0; j < NFR; ++j) {
              if (metric == PBBSS_PA_EUCLIDEAN) {
                // padded frames hold zeros on both sides
                const double d = x[b][j] - cent[a * TS + l + 16 * j];
                v[a * K + b] = fma(d, d, v[a * K + b]);
              } else {
                v[a * K + b] = fma(cent[a * TS + l + 16 * j], x[b][j], v[a * K + b]);
              }
            }
        row_reduce_scatter<NS>(v);
        const int e0 = QS * ((l >> 2) & 3);
        if ((l & 3) == 0 && f < end) {
#pragma unroll
          for (int r = 0; r < QS; ++r)
            if (e0 + r < KK) st_sc1(mine + K + (size_t)(f - start) * KK + e0 + r, v[r]);
        }
      };
#pragma unroll
      for (int p = 0; p < MAXP; ++p) {
        const int f = start + p * kSliceBins + rowid;
        if (p < npass) score_bin(f, fv[p]);
      }
      for (int p = MAXP; p < npass; ++p) {
        const int f = start + p * kSliceBins + rowid;
        double x[K][NFR];
        load_bin(f, f < end, x);
        score_bin(f, x);
      }
      // ---- the one exchange hop of the iteration
      team_barrier(ctrl, target, G, tid, spin_limit);
      gather(hop & 1, K + nb * KK);
      ++hop;
      __syncthreads();
      // ---- every workgroup solves every assignment of the segment (:342-350)
      int changed = 0;
      for (int i = tid; i < nb; i += kSliceThreads) {
        const int f = start + i;
        double sc[K][K];
        bool finite = true;
#pragma unroll
        for (int a = 0; a < K; ++a) {
          const double rn = cos ? 1.0 / fmax(sqrt(uni[a]), kTiny) : 1.0;
#pragma unroll
          for (int b = 0; b < K; ++b) {
            double x = uni[K + (size_t)i * KK + a * K + b];
            x = (metric == PBBSS_PA_EUCLIDEAN) ? -sqrt(x) : x * rn;  // :412-416
            finite = finite && isfinite(x);
            sc[a][b] = x;
          }
        }
        if (!finite) nonfinite = 1;  // reference: ValueError('score matrix is infeasible')
        int perm[K];
        assign_classes<K>(sc, optimal, perm);
        int pk = 0, nm = 0;
        const int mw = mapw[f];
        bool ident = true;
#pragma unroll
        for (int a = 0; a < K; ++a) {
          ident = ident && (perm[a] == a);
          pk |= perm[a] << (4 * a);
          nm |= ((mw >> (4 * perm[a])) & 15) << (4 * a);  // mapping[:, f] = mapping[perm, f]
        }
        lastp[f] = ident ? 0 : pk;
        if (!ident) {
          mapw[f] = nm;
          changed = 1;
        }
      }
      const int any = __syncthreads_or(changed);
      if constexpr (PROBE) {
        const int bad = __syncthreads_or(nonfinite);
        if ((any || bad) && tid == 0)
          __hip_atomic_store(my_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
      if (!any) break;  // nothing_changed (:352-353)
      // features[:, f, :] = features[perm, f, :] for the register-resident bins
#pragma unroll
      for (int p = 0; p < MAXP; ++p) {
        const int f = start + p * kSliceBins + rowid;
        const int lp = (f < end) ? lastp[f] : 0;
        if (lp != 0) {
          double y[K][NFR];
#pragma unroll
          for (int a = 0; a < K; ++a) {
            const int pa = (lp >> (4 * a)) & 15;
#pragma unroll
            for (int j = 0; j < NFR; ++j) {
              double x = fv[p][0][j];
#pragma unroll
              for (int k = 1; k < K; ++k) x = (pa == k) ? fv[p][k][j] : x;
              y[a][j] = x;
            }
          }
#pragma unroll
          for (int a = 0; a < K; ++a)
#pragma unroll
            for (int j = 0; j < NFR; ++j) fv[p][a][j] = y[a][j];
        }
      }
    }
  }
  if constexpr (PROBE) {
    if (tid == 0 && __hip_atomic_load(ctrl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
      __hip_atomic_store(my_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  // ---- output: the reverse mapping (dhtv_features_kernel then writes the aligned features over
  // the exchange area)
  if (g == 0)
    for (int i = tid; i < K * F; i += kSliceThreads) {
      const int a = i / F, f = i - a * F;
      mapping[i] = (mapw[f] >> (4 * a)) & 15;
    }
  const unsigned xerr = __hip_atomic_load(ctrl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if ((nonfinite || xerr) && lane == 0)
    atomicOr(status + u, (int32_t)(nonfinite ? PBBSS_ST_NONFINITE : 0) |
                             (int32_t)(xerr ? PBBSS_ST_EIG_NOCONV : 0));
}

// mask (U,K,F,T) gathered along the class axis: out[u,k,f,:] = mask[u,mapping[u,k,f],f,:]
__global__ void __launch_bounds__(256)
    apply_mapping_kernel(const double* __restrict__ mask, const int32_t* __restrict__ mapping,
                         int K, int F, int T, double* __restrict__ out) {
  const int64_t row = blockIdx.x;  // (u, k, f)
  const int64_t u = row / ((int64_t)K * F);
  const int f = (int)(row % F);
  const int src_k = mapping[row];
  const double* src = mask + ((u * K + src_k) * F + f) * (int64_t)T;
  double* dst = out + row * (int64_t)T;
  for (int t = threadIdx.x; t < T; t += blockDim.x) dst[t] = src[t];
}

// ---------------------------------------------------------------------------------------
// Pairwise solvers: OraclePermutationAlignment.calculate_mapping (permutation_alignment.py
// :703-786) and GreedyPermutationAlignment.calculate_mapping (:592-701).  Every bin is
// independent: one wavefront per (utterance, bin) forms the K x K score matrix of `mask`
// against `reference` (_ScoreMatrix.cos / multiply / euclidean, :396-417: rows = reference
// class, columns = mask class) and assigns the classes (:469-589).  The greedy solver's
// recursion mapping[:, f] = mapping[mapping[:, f-1], f] (:698-699) is a composition of
// permutations -- associative -- so it runs as one wave scan per utterance.
// ---------------------------------------------------------------------------------------
struct PaPairArgs {
  const double* mask;
  const double* ref;
  int64_t m_su, m_sk, m_sf;  // element strides of (utterance, class, bin); frames contiguous
  int64_t r_su, r_sk, r_sf;
  int64_t U, F;
  int T;
  int metric, optimal;
  double* scores;     // (U, F, K, K) or null
  int32_t* mapping;   // (U, K, map_F): column map_col0 + f receives the bin's permutation
  int64_t map_F, map_col0;
  int32_t* status;    // (U)
};

template <int K>
__global__ void __launch_bounds__(256) pa_pair_kernel(PaPairArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t id = (int64_t)blockIdx.x * 4 + wave;
  if (id >= a.U * a.F) return;
  const int64_t u = id / a.F, f = id - u * a.F;
  const double* m = a.mask + u * a.m_su + f * a.m_sf;
  const double* r = a.ref + u * a.r_su + f * a.r_sf;
  const int T = a.T;
  double nm[K], nr[K];
#pragma unroll
  for (int k = 0; k < K; ++k) nm[k] = nr[k] = 1.0;
  if (a.metric == PBBSS_PA_COS) {
    // _parameterized_vector_norm (:358-377): a / max(||a||, tiny) along the frames
    double sm[K], sr[K];
#pragma unroll
    for (int k = 0; k < K; ++k) sm[k] = sr[k] = 0.0;
    for (int t = lane; t < T; t += kWave) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        double x = m[k * a.m_sk + t], y = r[k * a.r_sk + t];
        sm[k] = fma(x, x, sm[k]);
        sr[k] = fma(y, y, sr[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      nm[k] = fmax(sqrt(wave_sum(sm[k])), kTiny);
      nr[k] = fmax(sqrt(wave_sum(sr[k])), kTiny);
    }
  }
  double sc[K][K];  // [reference class][mask class]
#pragma unroll
  for (int i = 0; i < K; ++i)
#pragma unroll
    for (int j = 0; j < K; ++j) sc[i][j] = 0.0;
  for (int t = lane; t < T; t += kWave) {
    double mv[K], rv[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      mv[k] = m[k * a.m_sk + t];
      rv[k] = r[k * a.r_sk + t];
    }
    if (a.metric == PBBSS_PA_COS) {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        mv[k] /= nm[k];
        rv[k] /= nr[k];
      }
    }
    if (a.metric == PBBSS_PA_EUCLIDEAN) {
#pragma unroll
      for (int i = 0; i < K; ++i)
#pragma unroll
        for (int j = 0; j < K; ++j) {
          double d = mv[j] - rv[i];
          sc[i][j] = fma(d, d, sc[i][j]);
        }
    } else {
#pragma unroll
      for (int i = 0; i < K; ++i)
#pragma unroll
        for (int j = 0; j < K; ++j) sc[i][j] = fma(rv[i], mv[j], sc[i][j]);
    }
  }
  bool finite = true;
#pragma unroll
  for (int i = 0; i < K; ++i)
#pragma unroll
    for (int j = 0; j < K; ++j) {
      double v = wave_sum(sc[i][j]);
      if (a.metric == PBBSS_PA_EUCLIDEAN) v = -sqrt(v);  // distance -> similarity (:412-416)
      sc[i][j] = v;
      finite = finite && isfinite(v);
      if (a.scores && lane == 0) a.scores[((u * a.F + f) * K + i) * K + j] = v;
    }
  if (!finite) {  // reference: ValueError('score matrix is infeasible') (:512-514)
    if (lane == 0) atomicOr(a.status + u, (int32_t)PBBSS_ST_NONFINITE);
  }
  int perm[K];
  assign_classes<K>(sc, a.optimal, perm);
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < K; ++i) a.mapping[(u * K + i) * a.map_F + a.map_col0 + f] = perm[i];
  }
}

// permutations of K <= 8 classes packed 4 bits per entry; (x o y)[k] = x[y[k]]
__device__ __forceinline__ unsigned perm_compose(unsigned x, unsigned y, int K) {
  unsigned r = 0;
  for (int k = 0; k < K; ++k) r |= ((x >> (4 * ((y >> (4 * k)) & 15u))) & 15u) << (4 * k);
  return r;
}

// Greedy solver, :690-699: column 0 = identity, then mapping[:, f] = mapping[mapping[:, f-1], f]
// for f = 1 .. F-1, i.e. new_f = M_f o M_{f-1} o ... o M_1.  One wavefront per utterance: each
// lane composes a contiguous chunk of bins, an inclusive wave scan combines the chunk
// aggregates, then the lane rewrites its chunk starting from its prefix.
__global__ void __launch_bounds__(64) pa_compose_kernel(int32_t* mapping, int K, int64_t F) {
  const int lane = threadIdx.x;
  int32_t* mp = mapping + (int64_t)blockIdx.x * K * F;
  unsigned ident = 0;
  for (int k = 0; k < K; ++k) ident |= (unsigned)k << (4 * k);
  if (lane == 0)
    for (int k = 0; k < K; ++k) mp[(int64_t)k * F] = k;
  const int64_t c = (F - 1 + 63) / 64;
  const int64_t f0 = 1 + lane * c, f1 = (f0 + c < F) ? f0 + c : F;
  unsigned agg = ident;
  for (int64_t f = f0; f < f1; ++f) {
    unsigned M = 0;
    for (int k = 0; k < K; ++k) M |= ((unsigned)mp[(int64_t)k * F + f] & 15u) << (4 * k);
    agg = perm_compose(M, agg, K);
  }
  unsigned x = agg;
  for (int d = 1; d < 64; d <<= 1) {
    unsigned o = __shfl_up(x, d, 64);
    if (lane >= d) x = perm_compose(x, o, K);
  }
  unsigned cur = __shfl_up(x, 1, 64);
  if (lane == 0) cur = ident;
  for (int64_t f = f0; f < f1; ++f) {
    unsigned M = 0;
    for (int k = 0; k < K; ++k) M |= ((unsigned)mp[(int64_t)k * F + f] & 15u) << (4 * k);
    cur = perm_compose(M, cur, K);
    for (int k = 0; k < K; ++k) mp[(int64_t)k * F + f] = (int32_t)((cur >> (4 * k)) & 15u);
  }
}

// _mapping_from_score_matrix (:469-589) on given scores (N, K, K): thread = matrix
template <int K>
__global__ void __launch_bounds__(256)
    pa_assign_kernel(const double* __restrict__ scores, int64_t N, int optimal,
                     int32_t* __restrict__ mapping, int32_t* __restrict__ status) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  double sc[K][K];
  bool finite = true;
#pragma unroll
  for (int i = 0; i < K; ++i)
#pragma unroll
    for (int j = 0; j < K; ++j) {
      sc[i][j] = scores[(n * K + i) * K + j];
      finite = finite && isfinite(sc[i][j]);
    }
  if (!finite) atomicOr(status, (int32_t)PBBSS_ST_NONFINITE);
  int perm[K];
  assign_classes<K>(sc, optimal, perm);
#pragma unroll
  for (int i = 0; i < K; ++i) mapping[(int64_t)i * N + n] = perm[i];
}

#define PBBSS_PA_SWITCH(K, CASE)                                                               \
  switch (K) {                                                                                 \
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8)                            \
    default: return PBBSS_ERR_UNSUPPORTED;                                                     \
  }

int launch_pa_pair(const double* mask, const double* ref, int64_t U, int K, int64_t F, int T,
                   const int64_t* ms, const int64_t* rs, int metric, int optimal, double* scores,
                   int32_t* mapping, int64_t map_F, int64_t map_col0, int32_t* status,
                   hipStream_t s) {
  if (K < 1 || K > kDhtvMaxK) return PBBSS_ERR_UNSUPPORTED;
  if (metric < PBBSS_PA_COS || metric > PBBSS_PA_EUCLIDEAN) return PBBSS_ERR_INVALID_ARG;
  PaPairArgs a{mask, ref, ms[0], ms[1], ms[2], rs[0], rs[1], rs[2], U, F, T, metric, optimal,
               scores, mapping, map_F, map_col0, status};
  const int64_t blocks = (U * F + 3) / 4;
  if (blocks > 2147483647LL) return PBBSS_ERR_UNSUPPORTED;
#define PBBSS_PA_CASE(KK)                                                                      \
  case KK:                                                                                     \
    hipLaunchKernelGGL(pa_pair_kernel<KK>, dim3((unsigned)blocks), dim3(256), 0, s, a);       \
    break;
  PBBSS_PA_SWITCH(K, PBBSS_PA_CASE)
#undef PBBSS_PA_CASE
  return hipGetLastError() == hipSuccess ? PBBSS_OK : PBBSS_ERR_HIP;
}

int launch_pa_compose(int32_t* mapping, int64_t U, int K, int64_t F, hipStream_t s) {
  if (K < 1 || K > kDhtvMaxK) return PBBSS_ERR_UNSUPPORTED;
  if (U > 2147483647LL) return PBBSS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(pa_compose_kernel, dim3((unsigned)U), dim3(64), 0, s, mapping, K, F);
  return hipGetLastError() == hipSuccess ? PBBSS_OK : PBBSS_ERR_HIP;
}

int launch_pa_assign(const double* scores, int64_t N, int K, int optimal, int32_t* mapping,
                     int32_t* status, hipStream_t s) {
  if (K < 1 || K > kDhtvMaxK) return PBBSS_ERR_UNSUPPORTED;
  const int64_t blocks = (N + 255) / 256;
  if (blocks > 2147483647LL) return PBBSS_ERR_UNSUPPORTED;
#define PBBSS_PA_CASE(KK)                                                                      \
  case KK:                                                                                     \
    hipLaunchKernelGGL(pa_assign_kernel<KK>, dim3((unsigned)blocks), dim3(256), 0, s, scores,  \
                       N, optimal, mapping, status);                                           \
    break;
  PBBSS_PA_SWITCH(K, PBBSS_PA_CASE)
#undef PBBSS_PA_CASE
  return hipGetLastError() == hipSuccess ? PBBSS_OK : PBBSS_ERR_HIP;
}
#undef PBBSS_PA_SWITCH

// frame-slice kernel: launches when the shape admits it (returns false otherwise)
template <int K, int NF>
static bool slice_launch_one(const double* mask, int64_t U, int F, int T, const int32_t* plan, int P,
                             int optimal, int metric, double* feat, int32_t* mapping,
                             int32_t* status, int G, size_t lds, unsigned* ctrl, double* scale,
                             bool probe, bool features, unsigned spin_limit, hipStream_t s, int* rc) {
  auto kfn = dhtv_slice_kernel<K, NF, false>;
  auto pfn = dhtv_slice_kernel<K, NF, true>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
      (probe && hipFuncSetAttribute(reinterpret_cast<const void*>(pfn),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
                    hipSuccess)) {
    *rc = PBBSS_ERR_HIP;
    return true;
  }
  // row scales on the whole chip -> [probe: every segment at once on P teams per utterance] ->
  // the plan on G workgroups per utterance -> aligned features
  // control words: [4 per utterance][probe: 4 + 1 flag per (utterance, segment)]
  const int64_t rows = U * K * F;
  unsigned* pctrl = ctrl + 4 * U;
  unsigned* pflag = probe ? pctrl + 4 * U * P : nullptr;
  const int nctrl = (int)(probe ? 4 * U + 5 * U * P : 4 * U);
  hipLaunchKernelGGL(dhtv_rowscale_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, mask,
                     rows, T, K * F, metric == PBBSS_PA_COS ? 1 : 0, scale, status, ctrl, nctrl);
  if (probe)
    hipLaunchKernelGGL(pfn, dim3((unsigned)(U * P * G)), dim3(kSliceThreads), lds, s, mask, scale,
                       feat, mapping, plan, P, F, T, optimal, metric, status, G, pctrl, pflag,
                       spin_limit);
  hipLaunchKernelGGL(kfn, dim3((unsigned)(U * G)), dim3(kSliceThreads), lds, s, mask, scale, feat,
                     mapping, plan, P, F, T, optimal, metric, status, G, ctrl, pflag, spin_limit);
  if (features)
    hipLaunchKernelGGL(dhtv_features_kernel, dim3((unsigned)rows), dim3(256), 0, s, mask, mapping,
                       scale, K, F, T, feat);
  *rc = hipGetLastError() == hipSuccess ? PBBSS_OK : PBBSS_ERR_HIP;
  return true;
}

static bool launch_slice(const double* mask, int64_t U, int K, int F, int T, const int32_t* plan,
                         int P, int optimal, int metric, double* feat, int32_t* mapping,
                         int32_t* status, size_t lds_limit, int num_cu, int want_team,
                         unsigned* ctrl, double* scale, bool want_probe, bool features,
                         unsigned spin_limit, hipStream_t s, int* rc) {
  if (K > 5) return false;  // K*K scores per bin live in registers through the butterfly
  if (U * K * F > 2147483647LL) return false;
  for (int NF = 4; NF <= 8; NF *= 2) {  // frames per lane; 16 * NF frames per workgroup
    if (K * NF > 24) break;
    const int G = (T + 16 * NF - 1) / (16 * NF);
    if (G < 2) break;
    if (want_team >= 2 && G > want_team && NF < 8 && K * NF * 2 <= 24) continue;
    // every team must be co-resident (one 512-thread workgroup per compute unit): leave half the
    // chip as margin once many teams are in flight -- a batch that large is served well enough
    // by one workgroup per utterance
    if ((int64_t)G * U > (U <= 4 ? num_cu : num_cu / 2)) continue;
    const size_t lds = slice_lds_bytes(K, NF, F);
    if (lds > lds_limit) continue;
    // exchange area inside the utterance's feature scratch
    if (2 * (size_t)G * slice_entries(K, F) > (size_t)K * F * T) continue;
    // the probe's P teams per utterance must be co-resident as well (one workgroup per CU)
    const bool probe = want_probe && P <= 64 && (int64_t)G * U * P <= num_cu;
#define PBBSS_SLICE_CASE(KK, NN)                                                               \
  if (K == KK && NF == NN)                                                                     \
    return slice_launch_one<KK, NN>(mask, U, F, T, plan, P, optimal, metric, feat, mapping,    \
                                    status, G, lds, ctrl, scale, probe, features, spin_limit, s, \
                                    rc);
    PBBSS_SLICE_CASE(1, 4) PBBSS_SLICE_CASE(1, 8)
    PBBSS_SLICE_CASE(2, 4) PBBSS_SLICE_CASE(2, 8)
    PBBSS_SLICE_CASE(3, 4) PBBSS_SLICE_CASE(3, 8)
    PBBSS_SLICE_CASE(4, 4)
    PBBSS_SLICE_CASE(5, 4)
#undef PBBSS_SLICE_CASE
  }
  return false;
}

// team_size: 0 automatic; 1 one workgroup per utterance; >= 2 frame-slice kernel with at most
// that many workgroups per utterance (as far as 256 frames per workgroup allow);
// <= -2 the bin-chunk team kernel with |team_size| workgroups (kept for A/B and for shapes the
// frame-slice kernel does not take: K > 5, fewer than 128 frames, no room for the exchange area).
int launch_dhtv(const double* mask, int64_t U, int K, int F, int T, const int32_t* plan, int P,
                int optimal, int metric, double* feat, int32_t* mapping, int32_t* status,
                size_t lds_limit, int num_cu, int team_size, void* team_buf, size_t team_bytes,
                int probe, unsigned spin_limit, hipStream_t s) {
  if (K < 1 || K > kDhtvMaxK) return PBBSS_ERR_UNSUPPORTED;
  if (metric < PBBSS_PA_COS || metric > PBBSS_PA_EUCLIDEAN) return PBBSS_ERR_INVALID_ARG;
  const size_t ctrl_bytes = (size_t)U * 4 * sizeof(unsigned);
  // the frame-slice path's probe keeps 4 words and a flag per (utterance, segment)
  const size_t ctrl_pad =
      (ctrl_bytes + (probe ? (size_t)U * P * 5 * sizeof(unsigned) : 0) + 255) & ~(size_t)255;
  unsigned* ctrl = static_cast<unsigned*>(team_buf);
  // team buffer: [control words][row scales of the frame-slice path]
  const size_t scale_bytes = (size_t)U * K * F * sizeof(double);
  if (team_buf && ctrl_pad + scale_bytes <= team_bytes && (team_size == 0 || team_size >= 2)) {
    int rc = PBBSS_OK;
    if (launch_slice(mask, U, K, F, T, plan, P, optimal, metric, feat, mapping, status, lds_limit,
                     num_cu, team_size, ctrl,
                     reinterpret_cast<double*>(static_cast<char*>(team_buf) + ctrl_pad),
                     (probe & 1) != 0, (probe & 2) == 0, spin_limit, s, &rc))
      return rc;
  }
  size_t lds = ((size_t)K * T + kDhtvWaves) * sizeof(double) + 16;
  if (lds > lds_limit) return PBBSS_ERR_LDS_CAPACITY;
  // team size: all U * G workgroups must be co-resident (one 1024-thread workgroup per CU)
  const int ncb = (K * T + kDhtvThreads - 1) / kDhtvThreads;
  int G = team_size < 0 ? -team_size : team_size;
  if (G <= 0) {  // default ~ 32 / sqrt(U): measured optimum 16 for 1..4 utterances, 8 for 16, 4 for 64
    G = 16;
    while (G > 2 && (int64_t)G * G * U > 1024) --G;
  }
  if (G > kDhtvTeamMax) G = kDhtvTeamMax;
  if ((int64_t)G * U > num_cu) G = (int)(num_cu / U);
  G = (G / ncb) * ncb;                                    // whole bin chunks per column block
  const size_t part_bytes = (size_t)U * (G > 0 ? G / ncb : 0) * K * T * sizeof(double);
  const bool team = team_buf && G >= 2 * ncb && ctrl_pad + part_bytes <= team_bytes;
  double* part = reinterpret_cast<double*>(static_cast<char*>(team_buf) + ctrl_pad);
  if (team && hipMemsetAsync(ctrl, 0, ctrl_bytes, s) != hipSuccess) return PBBSS_ERR_HIP;
#define PBBSS_DHTV_CASE(KK)                                                                     \
  case KK: {                                                                                    \
    if (team) {                                                                                 \
      auto kfn = dhtv_team_kernel<KK>;                                                          \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),                               \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
        return PBBSS_ERR_HIP;                                                                   \
      hipLaunchKernelGGL(kfn, dim3((unsigned)(U * G)), dim3(kDhtvThreads), lds, s, mask, feat,  \
                         mapping, plan, P, F, T, optimal, metric, status, G, part, ctrl, spin_limit);   \
    } else {                                                                                    \
      auto kfn = dhtv_kernel<KK>;                                                               \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),                               \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
        return PBBSS_ERR_HIP;                                                                   \
      hipLaunchKernelGGL(kfn, dim3((unsigned)U), dim3(kDhtvThreads), lds, s, mask, feat,        \
                         mapping, plan, P, F, T, optimal, metric, status);                        \
    }                                                                                           \
  } break;
  switch (K) {
    PBBSS_DHTV_CASE(1)
    PBBSS_DHTV_CASE(2)
    PBBSS_DHTV_CASE(3)
    PBBSS_DHTV_CASE(4)
    PBBSS_DHTV_CASE(5)
    PBBSS_DHTV_CASE(6)
    PBBSS_DHTV_CASE(7)
    PBBSS_DHTV_CASE(8)
    default: return PBBSS_ERR_UNSUPPORTED;
  }
#undef PBBSS_DHTV_CASE
  return hipGetLastError() == hipSuccess ? PBBSS_OK : PBBSS_ERR_HIP;
}

int launch_apply_mapping(const double* mask, const int32_t* mapping, int64_t U, int K, int F,
                         int T, double* out, hipStream_t s) {
  int64_t rows = U * K * F;
  if (rows > 2147483647LL) return PBBSS_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(apply_mapping_kernel, dim3((unsigned)rows), dim3(256), 0, s, mask, mapping, K,
                     F, T, out);
  return hipGetLastError() == hipSuccess ? PBBSS_OK : PBBSS_ERR_HIP;
}

}  // namespace pbbss
