// One translation unit per sensor count D (compiled with -DPBBSS_EM_D=<D>) so the
// heavy persistent-EM template instantiates in parallel under `make -j`.
#include "cacgmm_em.hpp"
#include "em_launch.hpp"

#ifndef PBBSS_EM_D
#error "compile with -DPBBSS_EM_D=<sensors>"
#endif

namespace pbbss {

template <int K, typename YS, bool SPILL>
static int launch_variant(EmArgs a, const EmLaunchCfg& cfg, hipStream_t stream) {
  using Kern = EmKernel<PBBSS_EM_D, K, YS, SPILL>;
  const size_t lds = Kern::lds_bytes(a.T);
  if (lds > cfg.lds_limit) return PBBSS_ERR_LDS_CAPACITY;
  auto kfn = cacgmm_em_kernel<PBBSS_EM_D, K, YS, SPILL>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return PBBSS_ERR_HIP;
  int occ = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kfn, kEmThreads, lds) != hipSuccess)
    return PBBSS_ERR_HIP;
  if (occ < 1) occ = 1;
  int64_t grid = (int64_t)cfg.num_cu * occ;
  if (grid > a.B) grid = a.B;
  if (SPILL) {
    a.scratch_stride = Kern::scratch_bytes(a.T);
    a.scratch = static_cast<char*>(cfg.get_scratch(cfg.scratch_ctx, a.scratch_stride * grid));
    if (!a.scratch) return PBBSS_ERR_HIP;
  }
  hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(kEmThreads), lds, stream, a);
  return hipGetLastError() == hipSuccess ? PBBSS_OK : PBBSS_ERR_HIP;
}

template <int K, typename YS>
static int launch_one(const EmArgs& a, const EmLaunchCfg& cfg, hipStream_t stream) {
  if (EmKernel<PBBSS_EM_D, K, YS, false>::lds_bytes(a.T) <= cfg.lds_limit)
    return launch_variant<K, YS, false>(a, cfg, stream);
  return launch_variant<K, YS, true>(a, cfg, stream);  // long utterance: frames in HBM/L2
}

template <typename YS>
static int launch_k(int K, const EmArgs& a, const EmLaunchCfg& cfg, hipStream_t stream) {
  switch (K) {
    case 1: return launch_one<1, YS>(a, cfg, stream);
    case 2: return launch_one<2, YS>(a, cfg, stream);
    case 3: return launch_one<3, YS>(a, cfg, stream);
    case 4: return launch_one<4, YS>(a, cfg, stream);
    case 5: return launch_one<5, YS>(a, cfg, stream);
    case 6: return launch_one<6, YS>(a, cfg, stream);
    default: return PBBSS_ERR_UNSUPPORTED;
  }
}

#define PBBSS_CAT2(a, b) a##b
#define PBBSS_CAT(a, b) PBBSS_CAT2(a, b)

int PBBSS_CAT(em_launch_d, PBBSS_EM_D)(int K, int y_is_c128, const EmArgs& a,
                                       const EmLaunchCfg& cfg, hipStream_t stream) {
  return y_is_c128 ? launch_k<double>(K, a, cfg, stream) : launch_k<float>(K, a, cfg, stream);
}

}  // namespace pbbss
