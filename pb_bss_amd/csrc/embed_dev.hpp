// Device code of the spectral half that more than one translation unit needs: the log-Bessel series
// of the von-Mises-Fisher normaliser and the IN-LAUNCH spectral finalize of the rotated joint loop
// (round 4).  Reference: distribution/von_mises_fisher.py:28-60 (log_norm), :122-144 (_fit),
// distribution/gaussian.py:152-193 (GaussianTrainer._fit, 'spherical').
#pragma once
#include <cmath>
#include "pbbss_dev.hpp"

namespace pbbss {

// ln(I_nu(x) / x^nu) by the ascending series, one wavefront per value (all 64 lanes call it)
__device__ inline double wave_log_bessel_over_power(double nu, double x, int lane) {
  const double q = 0.25 * x * x;
  const double lx = 2.0 * log(x) - 1.3862943611198906;  // ln(x^2 / 4)
  const int M = (int)ceil(fmin(x, 1.0e6)) + 48;        // terms fall by > 4x per step past m = x
  // lane owns the terms [m0, m0 + R): ln t_m0 from lgamma, then the block's sum RELATIVE to its
  // first term by the linear recurrence t_(m+1) / t_m = (x^2/4) / ((m+1)(m+1+nu)) -- no logarithm
  // or exponential per term (they were a serial chain of ~150 instructions per term on the one
  // wavefront the M-step finalize waits for).  Blocks longer than 16 terms (x > 1000) re-anchor
  // in the log domain so that the running product cannot overflow.
  const int R = (M + kWave - 1) / kWave;
  const int m0 = lane * R;
  double lt = (m0 ? (double)m0 * lx : 0.0) - lgamma((double)m0 + 1.0) - lgamma((double)m0 + nu + 1.0);
  double mx = lt, sum = 0.0;   // block total = exp(mx) * sum
  double p = 1.0, s = 1.0;     // running term and partial sum relative to the anchor term
  for (int r = 1; r < R; ++r) {
    const double m1 = (double)(m0 + r);
    p *= q / (m1 * (m1 + nu));
    s += p;
    if ((r & 15) == 15) {      // re-anchor: fold the partial sum into (mx, sum), restart at term r
      const double la = lt + log(s - p);  // the terms before r
      const double lp = lt + log(p);      // term r itself becomes the new anchor
      if (sum == 0.0) {
        mx = la;
        sum = 1.0;
      } else if (la > mx) {
        sum = sum * exp(mx - la) + 1.0;
        mx = la;
      } else {
        sum += exp(la - mx);
      }
      lt = lp;
      p = 1.0;
      s = 1.0;
    }
  }
  {
    const double la = lt + log(s);
    if (sum == 0.0) {
      mx = la;
      sum = 1.0;
    } else if (la > mx) {
      sum = sum * exp(mx - la) + 1.0;
      mx = la;
    } else {
      sum += exp(la - mx);
    }
  }
  const double gmx = wave_max(mx);
  sum = wave_sum(sum * exp(mx - gmx));
  return -nu * 0.6931471805599453 + gmx + log(sum);
}


// ---------------------------------------------------------------------------------------------
// Spectral finalize INSIDE the launch of the spatial kernel (cacgmm_em.hpp: run_joint_ms).
// The sweep (embed.hip: joint_sweep_kernel) leaves C chunk partials of W = K (E + 1) sums each
// (S1[k][e], S0[k]; the spherical Gaussian a second set S2'[k][e]).  Turning them into the model
// of the next sweep was a kernel of ONE workgroup between the sweep and the spatial kernel: 14-18
// us of every iteration on 1/256 of the chip.  The spatial kernel does not depend on it, so NH
// extra blocks of ITS grid do the work beside the bins: block h sums its share of the chunks
// (ascending order) into tmp[h][*], arrives on an agent-scope counter, and the last arriver adds
// the NH partial rows (ascending order again: bit-reproducible, no floating-point atomics),
// computes the model and puts the counter back to zero.
struct SpectralFin {
  int kind;           // PBBSS_EMBED_VMF / PBBSS_EMBED_GAUSS_SPHERICAL; < 0: no in-launch finalize
  int helpers;        // NH
  int C, E, K;
  const double* part; // [C][W] (+ [C][W] second moments right behind, Gaussian)
  double* tmp;        // [NH][2 W]
  unsigned* counter;  // zero between launches
  double cmin, cmax;  // vMF concentration clamp
  double* mean;       // (K, E)  in: previous mean (shift of the second moment), out: new mean
  double* scale;      // (K)     concentration / variance
  double* offset;     // (K)     log-pdf offsets of the next sweep
  double* prec;       // (K)
};

constexpr int kSpectralFinMaxW2 = 1024;  // values per helper thread loop bound (2 K (E + 1))

// called by all threads of helper block h (blockDim.x = nthreads, a multiple of 64); smem:
// (kSpectralFinMaxW2 + 1) doubles of LDS; 2 K (E + 1) <= kSpectralFinMaxW2.
__device__ inline void spectral_finalize_block(const SpectralFin& f, int h, int tid, int nthreads,
                                               double* smem) {
  const bool gauss = f.kind != 0 /* PBBSS_EMBED_VMF */;
  const int W = f.K * (f.E + 1);
  const int W2 = gauss ? 2 * W : W;
  const int c0 = (int)((int64_t)f.C * h / f.helpers), c1 = (int)((int64_t)f.C * (h + 1) / f.helpers);
  // level 1: this block's chunks, 16 loads in flight per thread
  for (int i = tid; i < W2; i += nthreads) {
    const double* p = f.part + (i < W ? (size_t)i : (size_t)f.C * W + (i - W));
    double t = 0.0;
    int c = c0;
    for (; c + 16 <= c1; c += 16) {
      double a[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) a[u] = p[(size_t)(c + u) * W];
#pragma unroll
      for (int u = 0; u < 16; ++u) t += a[u];
    }
    for (; c < c1; ++c) t += p[(size_t)c * W];
    __hip_atomic_store(f.tmp + (size_t)h * W2 + i, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // hand-off as in the split protocol of cacgmm_em.hpp: write-through stores, every storing wave
  // drains them, one lane arrives on the agent-scope counter
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int* last_flag = reinterpret_cast<int*>(smem + kSpectralFinMaxW2);
  if (tid == 0) {
    const unsigned before =
        __hip_atomic_fetch_add(f.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *last_flag = (before == (unsigned)(f.helpers - 1));
  }
  __syncthreads();
  if (!*last_flag) return;
  // level 2 (last arriver): totals into LDS
  for (int i = tid; i < W2; i += nthreads) {
    double t = 0.0;
    for (int hh = 0; hh < f.helpers; ++hh)
      t += __hip_atomic_load(f.tmp + (size_t)hh * W2 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    smem[i] = t;
  }
  if (tid == 0) __hip_atomic_store(f.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int lane = tid & (kWave - 1), wave = tid / kWave, nw = nthreads / kWave;
  const int E = f.E;
  for (int k = wave; k < f.K; k += nw) {
    const double* r1 = smem + k * (E + 1);
    const double s0 = r1[E];
    if (!gauss) {
      // Banerjee 2005 eq. 2.4 / 2.5 / 4.4 (von_mises_fisher.py:122-144)
      double n2 = 0.0;
      for (int d = lane; d < E; d += kWave) n2 = fma(r1[d], r1[d], n2);
      n2 = wave_sum(n2);
      const double norm = sqrt(n2);
      const double rn = 1.0 / fmax(norm, kTiny);
      for (int d = lane; d < E; d += kWave) f.mean[(size_t)k * E + d] = r1[d] * rn;
      const double rbar = norm / s0;
      double conc = (rbar * E - rbar * rbar * rbar) / (1.0 - rbar * rbar);
      conc = conc < f.cmin ? f.cmin : (conc > f.cmax ? f.cmax : conc);  // NaN stays NaN
      const double off = -(0.5 * E * 1.8378770664093454 +
                           wave_log_bessel_over_power(0.5 * E - 1.0, conc, lane));
      if (lane == 0) {
        f.scale[k] = conc;
        f.offset[k] = off;
        f.prec[k] = conc;
      }
    } else {
      // mean = S1 / den; variance about the NEW mean from the moment about the shift c (the
      // previous mean): sum w (y - m)^2 = S2' - 2 (m - c)(S1 - c S0) + (m - c)^2 S0 per dimension
      const double* r2 = smem + W + k * (E + 1);
      const double den = fmax(s0, kTiny);  // gaussian.py:160-163
      double acc = 0.0;
      for (int d = lane; d < E; d += kWave) {
        const double cshift = f.mean[(size_t)k * E + d];
        const double m = r1[d] / den;
        const double dm = m - cshift;
        acc += r2[d] - 2.0 * dm * (r1[d] - cshift * s0) + dm * dm * s0;
        f.mean[(size_t)k * E + d] = m;
      }
      acc = wave_sum(acc);
      if (lane == 0) {
        const double cv = acc / (den * (double)E);  // 'spherical', gaussian.py:179-182
        f.scale[k] = cv;
        const double pc = 1.0 / sqrt(cv);
        f.offset[k] = -0.5 * E * 1.8378770664093454 + (double)E * log(pc);
        f.prec[k] = pc;
      }
    }
  }
}

}  // namespace pbbss
