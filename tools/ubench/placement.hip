// Where does the dispatcher put the persistent EM workgroups?  Emulates the resource footprint of
// the headline launch -- 512 workgroups of 256 threads, 52.8 KB LDS, 168 VGPRs (3 fit a CU) --
// with and without 8 small "member" workgroups started first (same kernel or second stream),
// records the hardware id (XCC, SE, CU) of every workgroup and prints the histogram of
// workgroups per CU.   hipcc --offload-arch=gfx950 -O2 -o placement placement.hip && ./placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__device__ __forceinline__ unsigned hw_id() { return __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4); }
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20); }

__global__ void __launch_bounds__(256, 3) body(unsigned* ids, int offset, long long spin_cycles) {
  extern __shared__ char smem[];
  asm volatile("" ::: "v167");  // 168 VGPRs like the EM kernel
  if (threadIdx.x == 0) {
    ids[offset + blockIdx.x] = (xcc_id() << 16) | (hw_id() & 0xffff);
    smem[0] = 1;
  }
  const long long t0 = clock64();
  while (clock64() - t0 < spin_cycles) __builtin_amdgcn_s_sleep(8);
}

static void report(const char* tag, const std::vector<unsigned>& ids, int n_first, int n_main) {
  std::map<unsigned, int> per_cu_main, per_cu_first;
  for (int i = 0; i < n_first; ++i) per_cu_first[ids[i] & 0xffff3f00u | (ids[i] & 0xe000u)]++;
  for (int i = n_first; i < n_first + n_main; ++i) per_cu_main[ids[i] & 0xffff3f00u | (ids[i] & 0xe000u)]++;
  std::map<int, int> hist;
  for (auto& kv : per_cu_main) hist[kv.second]++;
  printf("%-34s CUs used by main: %zu; main workgroups per CU:", tag, per_cu_main.size());
  for (auto& kv : hist) printf("  %d x%d", kv.first, kv.second);
  int both = 0, max_with_member = 0;
  for (auto& kv : per_cu_first) {
    auto it = per_cu_main.find(kv.first);
    if (it != per_cu_main.end()) { both++; if (it->second > max_with_member) max_with_member = it->second; }
  }
  printf("   | member CUs %zu, of them hosting main %d (max main there %d)\n", per_cu_first.size(), both, max_with_member);
}

int main() {
  unsigned* d;
  hipMalloc(&d, 4096 * 4);
  std::vector<unsigned> h(4096);
  hipStream_t hi;
  int least, greatest;
  hipDeviceGetStreamPriorityRange(&least, &greatest);
  hipStreamCreateWithPriority(&hi, hipStreamNonBlocking, greatest);
  const long long spin = 2000000;  // ~1 ms
  const size_t lds_main = 52816, lds_small = 9000;
  hipFuncSetAttribute((const void*)body, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int rep = 0; rep < 2; ++rep) {
    hipMemset(d, 0, 4096 * 4);
    body<<<512, 256, lds_main>>>(d, 0, spin);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, 4096 * 4, hipMemcpyDeviceToHost);
    report("512 main alone", h, 0, 512);
    // members first on a second (high priority) stream, then main
    hipMemset(d, 0, 4096 * 4);
    body<<<8, 256, lds_small, hi>>>(d, 0, spin);
    body<<<512, 256, lds_main>>>(d, 8, spin);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, 4096 * 4, hipMemcpyDeviceToHost);
    report("8 members (2nd stream) + 512 main", h, 8, 512);
    // one grid: blocks 0..7 members (same LDS), then main
    hipMemset(d, 0, 4096 * 4);
    body<<<520, 256, lds_main>>>(d, 0, spin);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, 4096 * 4, hipMemcpyDeviceToHost);
    report("520 in one grid (first 8 = members)", h, 8, 512);
    hipMemset(d, 0, 4096 * 4);
    body<<<520, 256, 56000>>>(d, 0, spin);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, 4096 * 4, hipMemcpyDeviceToHost);
    report("520 in one grid, LDS 56000", h, 8, 512);
  }
  return 0;
}
