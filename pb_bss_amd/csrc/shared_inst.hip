// Persistent EM with mixture weights shared by a group of problems (weight_constant_axis
// (-3,) / (-3, -1) of CACGMMTrainer.fit, cacgmm.py:59, :142-157): one cooperative launch per
// batch of groups whose workgroups fit the device at once; the groups exchange their masked
// affiliations through agent-coherent memory once per iteration (EmKernel::run_shared).
// One translation unit per sensor count D, like em_inst.hip.
#include "cacgmm_em.hpp"
#include "em_launch.hpp"

#ifndef PBBSS_EM_D
#error "compile with -DPBBSS_EM_D=<sensors>"
#endif

namespace pbbss {

template <int K, typename YS>
static int launch_shared_one(EmArgs a, const EmLaunchCfg& cfg, hipStream_t stream) {
  using Kern = EmKernel<PBBSS_EM_D, K, YS, false>;
  const size_t lds = Kern::lds_bytes(a.T);
  if (lds > cfg.lds_limit) return PBBSS_ERR_UNSUPPORTED;  // frames must be LDS-resident
  auto kfn = cacgmm_em_shared_kernel<PBBSS_EM_D, K, YS>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return PBBSS_ERR_HIP;
  int occ = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kfn, kEmThreads, lds) != hipSuccess)
    return PBBSS_ERR_HIP;
  const int64_t capacity = (int64_t)cfg.num_cu * occ;
  if (a.wgroup > capacity) return PBBSS_ERR_UNSUPPORTED;  // a group must be co-resident
  const int64_t ngroups = a.B / a.wgroup;
  const int64_t per_launch = capacity / a.wgroup;

  // exchange workspace: counters | class sums | affiliations | reduced weights
  const int TS = a.T;
  const size_t n_cnt = (size_t)ngroups * 16 * sizeof(unsigned);
  const size_t head = (n_cnt + 255) & ~(size_t)255;
  const size_t n_gsum = (size_t)2 * a.B * K * sizeof(double);
  const bool kt = a.weight_mode == PBBSS_WEIGHT_SHARED_KT;
  const size_t n_gaff = kt ? (size_t)2 * a.B * K * TS * sizeof(double) : 0;
  const size_t n_gw = kt ? (size_t)2 * ngroups * K * TS * sizeof(double) : 0;
  char* ws = static_cast<char*>(cfg.get_scratch(cfg.scratch_ctx, head + n_gsum + n_gaff + n_gw));
  if (!ws) return PBBSS_ERR_HIP;
  a.gcount = reinterpret_cast<unsigned*>(ws);
  a.gsum = reinterpret_cast<double*>(ws + head);
  a.gaff = kt ? reinterpret_cast<double*>(ws + head + n_gsum) : nullptr;
  a.gw = kt ? reinterpret_cast<double*>(ws + head + n_gsum + n_gaff) : nullptr;
  a.xerror = reinterpret_cast<int*>(cfg.xbuf + 128);
  a.spin_limit = cfg.spin_limit;  // [0] this call, [16] sticky (pbbss_split_error)
  if (hipMemsetAsync(ws, 0, head, stream) != hipSuccess) return PBBSS_ERR_HIP;
  if (hipMemsetAsync(cfg.xbuf + 128, 0, 64, stream) != hipSuccess) return PBBSS_ERR_HIP;
  a.wb = a.wk = a.wt = 0;
  for (int64_t g0 = 0; g0 < ngroups; g0 += per_launch) {
    const int64_t ng = (ngroups - g0 < per_launch) ? ngroups - g0 : per_launch;
    a.b_first = g0 * a.wgroup;
    void* params[] = {&a};
    if (hipLaunchCooperativeKernel(reinterpret_cast<const void*>(kfn),
                                   dim3((unsigned)(ng * a.wgroup)), dim3(kEmThreads), params,
                                   (unsigned)lds, stream) != hipSuccess) {
      (void)hipGetLastError();
      return PBBSS_ERR_UNSUPPORTED;  // the caller falls back to the step-wise loop
    }
  }
  return PBBSS_OK;
}

template <typename YS>
static int launch_shared_k(int K, const EmArgs& a, const EmLaunchCfg& cfg, hipStream_t stream) {
  switch (K) {
    case 1: return launch_shared_one<1, YS>(a, cfg, stream);
    case 2: return launch_shared_one<2, YS>(a, cfg, stream);
    case 3: return launch_shared_one<3, YS>(a, cfg, stream);
    case 4: return launch_shared_one<4, YS>(a, cfg, stream);
    default: return PBBSS_ERR_UNSUPPORTED;
  }
}

#define PBBSS_CAT2(a, b) a##b
#define PBBSS_CAT(a, b) PBBSS_CAT2(a, b)

int PBBSS_CAT(em_shared_launch_d, PBBSS_EM_D)(int K, int y_is_c128, const EmArgs& a,
                                              const EmLaunchCfg& cfg, hipStream_t stream) {
  return y_is_c128 ? launch_shared_k<double>(K, a, cfg, stream)
                   : launch_shared_k<float>(K, a, cfg, stream);
}

}  // namespace pbbss
