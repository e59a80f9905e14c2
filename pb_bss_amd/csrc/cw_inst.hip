// Complex-Watson mixture EM kernels, one translation unit per sensor count D
// (compiled with -DPBBSS_EM_D=<D>), like em_inst.hip.
#include <cstdlib>
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

// Eight wavefronts per bin (cwmm.hpp: WatsonWide) for a launch in which every bin has a compute
// unit of its own: two wavefronts per SIMD instead of one.  PBBSS_ERR_UNSUPPORTED: not served.
template <int K, typename YS>
static int cw_launch_wide(const WatsonArgs& wa, const EmLaunchCfg& cfg, hipStream_t stream) {
  using Kern = WatsonWide<PBBSS_EM_D, K, YS>;
  static const bool off = [] {  // development knob: PBBSS_CW_WIDE=0 keeps the four-wave kernel
    const char* v = getenv("PBBSS_CW_WIDE");
    return v && v[0] == '0';
  }();
  const EmArgs& a = wa.em;
  if (off || a.B > cfg.num_cu || a.T <= 4 * kWave || a.wt != 0) return PBBSS_ERR_UNSUPPORTED;
  const size_t lds = Kern::lds_bytes(a.T);
  if (lds > cfg.lds_limit) return PBBSS_ERR_UNSUPPORTED;
  auto kfn = cwmm_em_wide_kernel<PBBSS_EM_D, K, YS>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return PBBSS_ERR_HIP;
  hipLaunchKernelGGL(kfn, dim3((unsigned)a.B), dim3(2 * kEmThreads), lds, stream, wa);
  return hipGetLastError() == hipSuccess ? PBBSS_OK : PBBSS_ERR_HIP;
}

// Remainder problems [b_first, b_first + r) as split groups on the side stream, concurrent with
// the main launch (em_inst.hip: launch_split; same exchange buffer, counters and epoch protocol).
template <int K, typename YS>
static int cw_launch_split(WatsonArgs wa, int64_t b_first, int r, const EmLaunchCfg& cfg) {
  using Base = EmKernel<PBBSS_EM_D, K, YS, false>;
  EmArgs& a = wa.em;
  const int window = cfg.split_window > 256 ? 256 : cfg.split_window;
  const int G = (a.T + window - 1) / window;
  const size_t lds = WatsonSplit<PBBSS_EM_D, K, YS>::lds_bytes(window);
  const size_t slab_bytes = Base::split_slab_doubles(r, G) * sizeof(double);
  const size_t head = 256;
  if (head + slab_bytes > cfg.xbuf_bytes) return PBBSS_ERR_UNSUPPORTED;
  auto kfn = cwmm_em_split_kernel<PBBSS_EM_D, K, YS>;
  if (!raise_lds_attribute(reinterpret_cast<const void*>(kfn), lds)) return PBBSS_ERR_HIP;
  a.T_total = a.T;
  a.split_groups = G;
  a.split_window = window;
  a.split_prio = cfg.split_prio;
  a.b_first = b_first;
  a.xcount = reinterpret_cast<unsigned*>(cfg.xbuf);
  a.xerror = reinterpret_cast<int*>(cfg.xbuf + 128);
  a.spin_limit = cfg.spin_limit;
  a.xslab = reinterpret_cast<double*>(cfg.xbuf + head);
  if (hipStreamWaitEvent(cfg.side_stream, cfg.ev_fork, 0) != hipSuccess) return PBBSS_ERR_HIP;
  a.xepoch = next_split_epoch(cfg);
  a.xbuf_given = (unsigned)cfg.xbuf_bytes;
  hipLaunchKernelGGL(kfn, dim3((unsigned)(r * G)), dim3(kEmThreads), lds, cfg.side_stream, wa);
  if (hipGetLastError() != hipSuccess) return PBBSS_ERR_HIP;
  if (hipEventRecord(cfg.ev_join, cfg.side_stream) != hipSuccess) return PBBSS_ERR_HIP;
  return PBBSS_OK;
}

// weight_constant_axis (-3, -1): one cooperative launch per batch of groups whose workgroups fit
// the device at once (shared_inst.hip: launch_shared_one is the cACGMM counterpart)
template <int K, typename YS>
static int cw_launch_shared(WatsonArgs wa, const EmLaunchCfg& cfg, hipStream_t stream) {
  using Kern = WatsonKernel<PBBSS_EM_D, K, YS, false>;
  EmArgs& a = wa.em;
  const size_t lds = Kern::lds_bytes(a.T);
  if (lds > cfg.lds_limit || !cfg.xbuf) return PBBSS_ERR_UNSUPPORTED;  // frames must be LDS-resident
  auto kfn = cwmm_em_shared_kernel<PBBSS_EM_D, K, YS>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return PBBSS_ERR_HIP;
  int occ = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kfn, kEmThreads, lds) != hipSuccess)
    return PBBSS_ERR_HIP;
  const int64_t capacity = (int64_t)cfg.num_cu * occ;
  if (a.wgroup < 1 || a.B % a.wgroup != 0 || a.wgroup > capacity) return PBBSS_ERR_UNSUPPORTED;
  const int64_t ngroups = a.B / a.wgroup;
  const int64_t per_launch = capacity / a.wgroup;
  const size_t n_cnt = (size_t)ngroups * 16 * sizeof(unsigned);
  const size_t head = (n_cnt + 255) & ~(size_t)255;
  const size_t n_gsum = (size_t)2 * a.B * K * sizeof(double);
  const bool kt = a.weight_mode == PBBSS_WEIGHT_SHARED_KT;
  const size_t n_gaff = kt ? (size_t)2 * a.B * K * a.T * sizeof(double) : 0;
  const size_t n_gw = kt ? (size_t)2 * ngroups * K * a.T * sizeof(double) : 0;
  char* ws = static_cast<char*>(cfg.get_scratch(cfg.scratch_ctx, head + n_gsum + n_gaff + n_gw));
  if (!ws) return PBBSS_ERR_HIP;
  a.gcount = reinterpret_cast<unsigned*>(ws);
  a.gsum = reinterpret_cast<double*>(ws + head);
  a.gaff = kt ? reinterpret_cast<double*>(ws + head + n_gsum) : nullptr;
  a.gw = kt ? reinterpret_cast<double*>(ws + head + n_gsum + n_gaff) : nullptr;
  a.xerror = reinterpret_cast<int*>(cfg.xbuf + 128);
  a.xepoch = 0;
  a.spin_limit = cfg.spin_limit;
  if (hipMemsetAsync(ws, 0, head, stream) != hipSuccess) return PBBSS_ERR_HIP;
  if (hipMemsetAsync(cfg.xbuf + 128, 0, 64, stream) != hipSuccess) return PBBSS_ERR_HIP;
  for (int64_t g0 = 0; g0 < ngroups; g0 += per_launch) {
    const int64_t ng = (ngroups - g0 < per_launch) ? ngroups - g0 : per_launch;
    a.b_first = g0 * a.wgroup;
    void* params[] = {&wa};
    if (hipLaunchCooperativeKernel(reinterpret_cast<const void*>(kfn),
                                   dim3((unsigned)(ng * a.wgroup)), dim3(kEmThreads), params,
                                   (unsigned)lds, stream) != hipSuccess) {
      (void)hipGetLastError();
      return PBBSS_ERR_UNSUPPORTED;  // the caller falls back to the step-wise loop
    }
  }
  return PBBSS_OK;
}

template <int K, typename YS>
static int cw_launch_one(const WatsonArgs& wa, const EmLaunchCfg& cfg, hipStream_t stream) {
  if (wa.em.weight_mode == PBBSS_WEIGHT_SHARED_K || wa.em.weight_mode == PBBSS_WEIGHT_SHARED_KT)
    return cw_launch_shared<K, YS>(wa, cfg, stream);
  if (WatsonKernel<PBBSS_EM_D, K, YS, false>::lds_bytes(wa.em.T) > cfg.lds_limit)
    return cw_launch_variant<K, YS, true>(wa, cfg, stream);
  // 2^n + 1 bins: the r remainder problems would put one more full workgroup on r compute units
  // and set the kernel time (257 bins: +24 %); G small member workgroups per problem instead
  const EmArgs& a = wa.em;
  const int64_t r = a.B % cfg.num_cu;
  const int window = cfg.split_window > 256 ? 256 : cfg.split_window;
  const size_t slab_need =
      256 + EmKernel<PBBSS_EM_D, K, YS, false>::split_slab_doubles((int)r, (a.T + window - 1) / window) *
                sizeof(double);
  const bool split = cfg.allow_split && cfg.side_stream && cfg.xbuf && a.iterations >= kSplitMinIterations &&
                     a.B > cfg.num_cu && a.B <= 3 * (int64_t)cfg.num_cu && r >= 1 &&
                     r <= kSplitMaxProblems && a.T >= 2 * cfg.split_window && a.wt == 0 &&
                     slab_need <= cfg.xbuf_bytes;
  if (!split) {
    const int rcw = cw_launch_wide<K, YS>(wa, cfg, stream);
    if (rcw != PBBSS_ERR_UNSUPPORTED) return rcw;
    return cw_launch_variant<K, YS, false>(wa, cfg, stream);
  }
  WatsonArgs main_wa = wa;
  main_wa.em.B = a.B - r;
  if (hipEventRecord(cfg.ev_fork, stream) != hipSuccess) return PBBSS_ERR_HIP;
  int rc = cw_launch_wide<K, YS>(main_wa, cfg, stream);
  if (rc == PBBSS_ERR_UNSUPPORTED) rc = cw_launch_variant<K, YS, false>(main_wa, cfg, stream);
  if (rc != PBBSS_OK) return rc;
  rc = cw_launch_split<K, YS>(wa, a.B - r, (int)r, cfg);
  if (rc != PBBSS_OK) return rc;
  if (hipStreamWaitEvent(stream, cfg.ev_join, 0) != hipSuccess) return PBBSS_ERR_HIP;
  return PBBSS_OK;
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
