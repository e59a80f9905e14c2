// Micro-benchmark: cost of cross-lane primitives on gfx950 (cycles per op, one wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../pb_bss_amd/csrc/pbbss_dev.hpp"
using namespace pbbss;

template <int MODE>
__global__ void k(double* out, unsigned long long* cyc) {
  double v[16];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1.5 + i;
  unsigned long long t0 = __builtin_readcyclecounter();
  asm volatile("" ::: "memory");
  for (int rep = 0; rep < 64; ++rep) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      double lo = v[i], hi = v[i + 8];
      if (MODE == 0) { swap32_f64(lo, hi); v[i] = lo + hi; v[i + 8] = hi; }
      if (MODE == 1) { swap16_f64(lo, hi); v[i] = lo + hi; v[i + 8] = hi; }
      if (MODE == 2) { v[i] = lo + dpp_f64<kDppRowRor8, 0xF>(lo, hi); }
      if (MODE == 3) { v[i] = lo + __shfl_xor(hi, 32, 64); }
      if (MODE == 4) { v[i] = lo + hi; }
      if (MODE == 5) { v[i] = lo + dpp_f64<kDppQuadXor1, 0xF>(lo, hi); }
    }
    asm volatile("" ::: "memory");
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int i = 0; i < 16; ++i) s += v[i];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[MODE] = t1 - t0;
}

int main() {
  double* out; unsigned long long* cyc;
  hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8 * 8);
  hipMemset(cyc, 0, 64);
  k<0><<<1, 64>>>(out, cyc); k<1><<<1, 64>>>(out, cyc); k<2><<<1, 64>>>(out, cyc);
  k<3><<<1, 64>>>(out, cyc); k<4><<<1, 64>>>(out, cyc); k<5><<<1, 64>>>(out, cyc);
  k<0><<<1, 64>>>(out, cyc); k<1><<<1, 64>>>(out, cyc); k<2><<<1, 64>>>(out, cyc);
  k<3><<<1, 64>>>(out, cyc); k<4><<<1, 64>>>(out, cyc); k<5><<<1, 64>>>(out, cyc);
  hipDeviceSynchronize();
  unsigned long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  const char* names[] = {"permlane32_swap x2 + add", "permlane16_swap x2 + add", "dpp row_ror x2 + add",
                         "shfl_xor(bpermute) x2 + add", "add only", "dpp quad_perm x2 + add"};
  for (int m = 0; m < 6; ++m) printf("%-30s %6.1f ticks per (exchange+add) [512 per run]\n", names[m], h[m] / 512.0);
  return 0;
}
