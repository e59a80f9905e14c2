// Complex-Watson mixture EM kernels, one translation unit per sensor count D
// (compiled with -DPBBSS_EM_D=<D>), like em_inst.hip.
#include "cwmm.hpp"
#include "em_launch.hpp"

#ifndef PBBSS_EM_D
#error "compile with -DPBBSS_EM_D=<sensors>"
#endif

namespace pbbss {

template <int K, typename YS, bool SPILL>
static int cw_launch_variant(WatsonArgs wa, const EmLaunchCfg& cfg, hipStream_t stream) {
  using Kern = WatsonKernel<PBBSS_EM_D, K, YS, SPILL>;
  const size_t lds = Kern::lds_bytes(wa.em.T);
  if (lds > cfg.lds_limit) return PBBSS_ERR_LDS_CAPACITY;
  auto kfn = cwmm_em_kernel<PBBSS_EM_D, K, YS, SPILL>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return PBBSS_ERR_HIP;
  int occ = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kfn, kEmThreads, lds) != hipSuccess)
    return PBBSS_ERR_HIP;
  if (occ < 1) occ = 1;
  int64_t grid = (int64_t)cfg.num_cu * occ;
  if (grid > wa.em.B) grid = wa.em.B;
  if (SPILL) {
    wa.em.scratch_stride = Kern::Base::scratch_bytes(wa.em.T);
    wa.em.scratch =
        static_cast<char*>(cfg.get_scratch(cfg.scratch_ctx, wa.em.scratch_stride * grid));
    if (!wa.em.scratch) return PBBSS_ERR_HIP;
  }
  hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(kEmThreads), lds, stream, wa);
  return hipGetLastError() == hipSuccess ? PBBSS_OK : PBBSS_ERR_HIP;
}

template <int K, typename YS>
static int cw_launch_one(const WatsonArgs& wa, const EmLaunchCfg& cfg, hipStream_t stream) {
  if (WatsonKernel<PBBSS_EM_D, K, YS, false>::lds_bytes(wa.em.T) <= cfg.lds_limit)
    return cw_launch_variant<K, YS, false>(wa, cfg, stream);
  return cw_launch_variant<K, YS, true>(wa, cfg, stream);
}

template <typename YS>
static int cw_launch_k(int K, const WatsonArgs& wa, const EmLaunchCfg& cfg, hipStream_t stream) {
  switch (K) {
    case 1: return cw_launch_one<1, YS>(wa, cfg, stream);
    case 2: return cw_launch_one<2, YS>(wa, cfg, stream);
    case 3: return cw_launch_one<3, YS>(wa, cfg, stream);
    case 4: return cw_launch_one<4, YS>(wa, cfg, stream);
    default: return PBBSS_ERR_UNSUPPORTED;
  }
}

#define PBBSS_CAT2(a, b) a##b
#define PBBSS_CAT(a, b) PBBSS_CAT2(a, b)

int PBBSS_CAT(cw_launch_d, PBBSS_EM_D)(int K, int y_is_c128, const WatsonArgs& wa,
                                       const EmLaunchCfg& cfg, hipStream_t stream) {
  return y_is_c128 ? cw_launch_k<double>(K, wa, cfg, stream)
                   : cw_launch_k<float>(K, wa, cfg, stream);
}

}  // namespace pbbss
