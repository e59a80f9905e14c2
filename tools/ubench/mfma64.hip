// Micro-benchmark: does v_mfma_f64_16x16x4_f64 on gfx950 run CONCURRENTLY with float64 VALU
// work of another wave on the same SIMD, and what is the accumulator layout?
//   hipcc --offload-arch=gfx950 -O3 -o mfma64 mfma64.hip && ./mfma64
// One 512-thread workgroup = 8 waves = 2 per SIMD.  Waves 0-3 run an MFMA loop (3 independent
// accumulators, like one class each), waves 4-7 a v_fma_f64 loop (48 independent chains).
// mode 1: MFMA waves only; mode 2: VALU waves only; mode 3: both.  If t(3) ~ max(t(1), t(2)) the
// two pipes overlap; if t(3) ~ t(1) + t(2) they share the float64 datapath.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(512) probe(double* out, int mode, int n_mfma, int n_valu) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  double sink = 0.0;
  if (wave < 4) {
    if (mode & 1) {
      double4_t c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0}, c2 = {0, 0, 0, 0};
      double a = 1.0 + lane * 1e-3, b = 1.0 - lane * 1e-3;
      for (int i = 0; i < n_mfma; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, c2, 0, 0, 0);
      }
      sink = c0[0] + c0[1] + c0[2] + c0[3] + c1[0] + c1[3] + c2[1] + c2[2];
    }
  } else {
    if (mode & 2) {
      double acc[48];
#pragma unroll
      for (int x = 0; x < 48; ++x) acc[x] = lane + x;
      double m = 1.0 + 1e-9 * lane, ad = 1e-7;
      for (int i = 0; i < n_valu; ++i) {
#pragma unroll
        for (int x = 0; x < 48; ++x) acc[x] = fma(acc[x], m, ad);
      }
#pragma unroll
      for (int x = 0; x < 48; ++x) sink += acc[x];
    }
  }
  out[(size_t)blockIdx.x * 512 + threadIdx.x] = sink;
}

// layout probe: A[i][k] = i+1 (k == 0), B[k][j] = j+1 (k == 0)  =>  D[i][j] = (i+1)(j+1)
__global__ void layout(double* out) {
  const int lane = threadIdx.x;
  double a = (lane / 16 == 0) ? (double)(lane % 16 + 1) : 0.0;
  double b = (lane / 16 == 0) ? (double)(lane % 16 + 1) : 0.0;
  double4_t c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];
}

static float run(double* out, int blocks, int mode, int nm, int nv) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  probe<<<blocks, 512>>>(out, mode, nm, nv);  // warm
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) probe<<<blocks, 512>>>(out, mode, nm, nv);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main() {
  double* out;
  hipMalloc(&out, (size_t)2048 * 512 * 8);
  layout<<<1, 64>>>(out);
  double h[256];
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  int ok_a = 1, ok_b = 1;  // a: row = 4r + lane/16 ; b: row = 4(lane/16) + r
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      int col = l % 16;
      int row = (int)(h[l * 4 + r] / (col + 1) + 0.5) - 1;
      if (row != 4 * r + l / 16) ok_a = 0;
      if (row != 4 * (l / 16) + r) ok_b = 0;
    }
  printf("layout: row=4r+lane/16: %s   row=4(lane/16)+r: %s   (col = lane%%16)\n", ok_a ? "YES" : "no",
         ok_b ? "YES" : "no");
  printf("lane 17 regs: %g %g %g %g (col 1 => rows = v/2-1)\n", h[68], h[69], h[70], h[71]);
  const int nm = 4000, nv = 4000;  // 12000 MFMA / 192000 FMA per wave
  for (int blocks : {1, 256, 512}) {
    float t1 = run(out, blocks, 1, nm, nv), t2 = run(out, blocks, 2, nm, nv), t3 = run(out, blocks, 3, nm, nv);
    printf("blocks %4d: mfma-only %.3f ms (%.1f ns/MFMA)  valu-only %.3f ms (%.2f ns/FMA)  both %.3f ms  "
           "=> both/max %.2f, both/sum %.2f\n",
           blocks, t1, t1 * 1e6 / (3.0 * nm), t2, t2 * 1e6 / (48.0 * nv), t3,
           t3 / (t1 > t2 ? t1 : t2), t3 / (t1 + t2));
  }
  // balance the two so neither hides trivially: scale valu work to ~ mfma time
  {
    float t1 = run(out, 512, 1, nm, nv), t2 = run(out, 512, 2, nm, nv);
    int nv2 = (int)(nv * t1 / t2);
    float t2b = run(out, 512, 2, nm, nv2), t3b = run(out, 512, 3, nm, nv2);
    printf("balanced (512 blocks, n_valu %d): mfma %.3f  valu %.3f  both %.3f  both/max %.2f both/sum %.2f\n",
           nv2, t1, t2b, t3b, t3b / (t1 > t2b ? t1 : t2b), t3b / (t1 + t2b));
  }
  return 0;
}
