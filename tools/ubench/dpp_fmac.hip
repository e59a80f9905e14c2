// Micro-benchmark: v_fmac_f64_dpp ... row_newbcast:N on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -o dpp_fmac dpp_fmac.hip && ./dpp_fmac
// DP-ALU DPP lets a float64 FMA take src0 from lane N of its own row of 16 lanes: one VGPR pair
// then carries 16 wave-uniform operands (replicated in the 4 rows) -- the alternative to an LDS
// broadcast read (8 LDS cycles per 16 bytes and wave) or an SGPR (s_load latency, <= 100 SGPRs)
// per operand.  Questions: (1) is the value what the ISA text says, (2) does the DPP form issue
// at the plain v_fmac_f64 rate, (3) with the operand registers refilled by ds_read_b64 at the
// rate a quadratic form needs (one 16-operand register per 16 FMAs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define FM(acc, a, y, n) \
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #n " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(y))
#define FP(acc, a, y) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(y))

__global__ void check(double* out) {
  const int lane = threadIdx.x;
  double a = 100.0 * (lane / 16) + (lane % 16);  // lane l of row r holds 100 r + l
  double y = 1.0, acc = 0.0;
  FM(acc, a, y, 5);
  out[lane] = acc;  // expect 100 r + 5
}

// mode 0: plain fmac, mode 1: DPP fmac, mode 2: DPP fmac + one ds_read_b64 refill per 16 FMAs
template <int MODE>
__global__ void __launch_bounds__(256) probe(double* out, int n) {
  __shared__ double lds[4096];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 1.0 + 1e-9 * i;
  __syncthreads();
  double acc[8], y[8];
#pragma unroll
  for (int x = 0; x < 8; ++x) { acc[x] = lane + x; y[x] = 1e-7 * (x + 1); }
  double a0 = 1.0 + 1e-9 * lane, a1 = 1.0 - 1e-9 * lane;
  const double* src = lds + (lane & 15);
  for (int i = 0; i < n; ++i) {
    if (MODE == 2) { a0 = src[(i & 127) * 32]; a1 = src[(i & 127) * 32 + 16]; }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (MODE == 0) {
        FP(acc[0], a0, y[0]); FP(acc[1], a0, y[1]); FP(acc[2], a0, y[2]); FP(acc[3], a0, y[3]);
        FP(acc[4], a1, y[4]); FP(acc[5], a1, y[5]); FP(acc[6], a1, y[6]); FP(acc[7], a1, y[7]);
        FP(acc[0], a1, y[7]); FP(acc[1], a1, y[6]); FP(acc[2], a1, y[5]); FP(acc[3], a1, y[4]);
        FP(acc[4], a0, y[3]); FP(acc[5], a0, y[2]); FP(acc[6], a0, y[1]); FP(acc[7], a0, y[0]);
      } else {
        FM(acc[0], a0, y[0], 0); FM(acc[1], a0, y[1], 1); FM(acc[2], a0, y[2], 2); FM(acc[3], a0, y[3], 3);
        FM(acc[4], a0, y[4], 4); FM(acc[5], a0, y[5], 5); FM(acc[6], a0, y[6], 6); FM(acc[7], a0, y[7], 7);
        FM(acc[0], a1, y[7], 8); FM(acc[1], a1, y[6], 9); FM(acc[2], a1, y[5], 10); FM(acc[3], a1, y[4], 11);
        FM(acc[4], a1, y[3], 12); FM(acc[5], a1, y[2], 13); FM(acc[6], a1, y[1], 14); FM(acc[7], a1, y[0], 15);
      }
    }
  }
  double s = 0.0;
#pragma unroll
  for (int x = 0; x < 8; ++x) s += acc[x];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
static float run(double* out, int blocks, int n) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  probe<MODE><<<blocks, 256>>>(out, n);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<MODE><<<blocks, 256>>>(out, n);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  double* out;
  hipMalloc(&out, sizeof(double) * 256 * 4096);
  check<<<1, 64>>>(out);
  double h[64];
  hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) bad += (h[l] != 100.0 * (l / 16) + 5);
  printf("row_newbcast:5 semantics (lane l of row r reads lane 5 of row r): %s  [%g %g %g %g]\n",
         bad ? "MISMATCH" : "ok", h[0], h[17], h[40], h[63]);
  const int n = 20000;  // x 32 FMAs
  for (int wg_per_cu = 1; wg_per_cu <= 2; ++wg_per_cu) {
    const int blocks = 256 * wg_per_cu;
    float t0 = run<0>(out, blocks, n), t1 = run<1>(out, blocks, n), t2 = run<2>(out, blocks, n);
    const double fm = (double)n * 32;
    printf("%d wave(s) per SIMD: plain %.3f ms (%.2f ns/FMA/wave)  dpp %.3f ms (%.2f)  dpp+ds_read refill %.3f ms (%.2f)\n",
           wg_per_cu, t0, t0 * 1e6 / fm / wg_per_cu, t1, t1 * 1e6 / fm / wg_per_cu, t2,
           t2 * 1e6 / fm / wg_per_cu);
  }
  return 0;
}
