// Micro-benchmark of the F phase (EmKernel::factor_class, fast path) in isolation:
// one workgroup, three waves factor three fixed HPD matrices `iters` times; prints shader
// cycles per call and, by re-running with 1/2/3 active waves, what is latency and what is issue.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ipb_bss_amd/csrc -Iinclude tools/ubench/fphase.hip -o gpurun_out/fphase && gpurun_out/fphase
#include <hip/hip_runtime.h>
#include <cstdio>
#include "cacgmm_em.hpp"
using namespace pbbss;
using Kern = EmKernel<8, 3, float, false>;

// mode 1: phase E (all four waves), mode 2: phase M, on a synthetic observation
__global__ void __launch_bounds__(256, 3) emprobe(EmArgs a, int mode, unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  auto L = Kern::carve(smem, a.T);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int t = tid; t < L.Tp; t += 256) {
    for (int dp = 0; dp < 4; ++dp) {
      float4 v = {0.1f * (t % 7) + dp, 0.2f - 0.01f * (t % 5), 0.3f + dp, -0.1f * (t % 3)};
      *reinterpret_cast<float4*>(L.ybuf + Kern::yoff(dp, t)) = v;  // 64-frame chunk layout (round 5)
    }
    L.inv_n2[t] = 0.05;
    for (int k = 0; k < 3; ++k) L.wbuf[Kern::woff(k, t)] = 0.3 + 0.1 * k;
  }
  for (int i = tid; i < 3 * 64; i += 256) L.apack[i] = (i % 64 < 8) ? 2.0 : 0.01 * (i % 5);
  if (tid < 3) {
    L.detm[tid] = 0.7;
    L.rdet[tid] = 1.0 / 0.7;
    L.dete[tid] = 1;
    L.wgt[tid] = 1.0 / 3;
  }
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < a.iterations; ++it) {
    if (mode == 1) {
      Kern::phase_e<false, false>(a, L, 0, tid, wave, lane, a.aff_eps);
    } else {
      switch (wave) {
        case 0: Kern::phase_m<0>(a, L, lane); break;
        case 1: Kern::phase_m<1>(a, L, lane); break;
        case 2: Kern::phase_m<2>(a, L, lane); break;
        default: Kern::phase_m<3>(a, L, lane); break;
      }
    }
    __syncthreads();
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (tid == 0) out[0] = t1 - t0;
}

__global__ void __launch_bounds__(256, 3) fprobe(EmArgs a, int active_waves, unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  auto L = Kern::carve(smem, a.T);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave < 3) {  // C_k = Hermitian, diagonally dominant
    const int i = lane >> 3, j = lane & 7;
    double re = (i == j) ? 3.0 + 0.1 * i + wave : 0.05 * (i + j + 1) / (1.0 + (i > j ? i - j : j - i));
    double im = (i == j) ? 0.0 : (i < j ? 0.02 * (j - i) : -0.02 * (i - j));
    // packed covariance sums as the M phase leaves them: diag i -> i, pair (i < j) -> D + 2 p + {Re, Im}
    if (i == j) {
      L.cpack[wave * Kern::NA + i] = re;
    } else if (i < j) {
      const int p = Kern::pair_index(i, j);
      L.cpack[wave * Kern::NA + 8 + 2 * p] = re;
      L.cpack[wave * Kern::NA + 8 + 2 * p + 1] = im;
    }
  }
  if (threadIdx.x < 12) L.red[threadIdx.x] = 40.0 + threadIdx.x;
  if (threadIdx.x < 3) L.status[threadIdx.x] = 0;
  if (threadIdx.x == 0) *L.flags = 0;
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < a.iterations; ++it) {
    if (wave < active_waves) Kern::factor_class(a, L, blockIdx.x, wave, lane, false);
    __syncthreads();
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) {
    out[0] = t1 - t0;
    out[1] = (unsigned long long)L.status[0];
    out[2] = (unsigned long long)(L.apack[0] * 1e6);
  }
}

int main() {
  EmArgs a{};
  a.T = 500;
  a.B = 1;
  a.iterations = 2000;
  a.covariance_norm = PBBSS_COVNORM_EIGENVALUE;
  a.weight_mode = PBBSS_WEIGHT_PER_CLASS_MEAN;
  a.eig_floor = 1e-10;
  unsigned long long* out;
  hipMalloc(&out, 64);
  const size_t lds = Kern::lds_bytes(a.T);
  hipFuncSetAttribute((const void*)fprobe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int w = 1; w <= 3; ++w) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(fprobe, dim3(1), dim3(256), lds, 0, a, w, out);
      unsigned long long h[3];
      hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
      if (rep) printf("active waves %d: %.0f cycles per F call (status %llu, apack0 %.6f)\n", w,
                      (double)h[0] / a.iterations, h[1], (double)h[2] * 1e-6);
    }
  }
  a.aff_eps = 1e-10;
  hipFuncSetAttribute((const void*)emprobe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int mode = 1; mode <= 2; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(emprobe, dim3(1), dim3(256), lds, 0, a, mode, out);
      unsigned long long h[1];
      hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
      if (rep) printf("phase %s: %.0f cycles per call (one workgroup, T = %d)\n", mode == 1 ? "E" : "M",
                      (double)h[0] / a.iterations, a.T);
    }
  }
  return 0;
}
