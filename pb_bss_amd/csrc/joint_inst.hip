// Joint spatial+spectral E/M step of the cACG half (gcacgmm.py / vmfcacgmm.py), one
// translation unit per sensor count D (-DPBBSS_EM_D=<D>) like em_inst.hip.
#include <cstdlib>
#include "cacgmm_em.hpp"
#include "em_launch.hpp"

#ifndef PBBSS_EM_D
#error "compile with -DPBBSS_EM_D=<sensors>"
#endif

namespace pbbss {

template <int K, typename YS>
static int launch_joint(const EmArgs& a, const JointExtras& jx, int inline_pa,
                        const EmLaunchCfg& cfg, hipStream_t stream) {
  using Kern = EmKernel<PBBSS_EM_D, K, YS, false>;
  size_t lds = Kern::lds_bytes(a.T) + 64;  // + class permutation of the inline PA
  // development knob: request at least this much LDS (e.g. 56000 pins two workgroups per CU)
  static const size_t lds_pad = [] {
    const char* v = getenv("PBBSS_JOINT_LDS_PAD");
    return v ? (size_t)atol(v) : (size_t)0;
  }();
  if (lds < lds_pad) lds = lds_pad;
  if (lds > cfg.lds_limit) return PBBSS_ERR_LDS_CAPACITY;
  auto kfn = inline_pa ? cacgmm_joint_kernel<PBBSS_EM_D, K, YS, true>
                       : cacgmm_joint_kernel<PBBSS_EM_D, K, YS, false>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return PBBSS_ERR_HIP;
  int occ = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kfn, kEmThreads, lds) != hipSuccess)
    return PBBSS_ERR_HIP;
  if (occ < 1) occ = 1;
  // Tail handling (the rule of launch_one in em_inst.hip): with B = m * num_cu + r (small r) the
  // r extra problems would put one more full workgroup on r CUs and set the time of EVERY
  // one-iteration launch; they run as member workgroups on frame windows instead
  // (run_joint_member; no spinning, the last arriver finishes the problem).
  const int64_t r = a.B % cfg.num_cu;
  const int window = cfg.split_window > 256 ? 256 : cfg.split_window;
  const int G = (a.T + window - 1) / window;
  const size_t slab_need = 256 + (size_t)r * G * Kern::kSlabLen * sizeof(double);
  const bool members = cfg.allow_split && !inline_pa && occ >= 3 && a.B > cfg.num_cu &&
                       a.B <= 2 * (int64_t)cfg.num_cu + kSplitMaxProblems && r >= 1 &&
                       r <= kSplitMaxProblems && a.T >= 2 * window && slab_need <= cfg.xbuf_bytes;
  if (members) {  // pbbss_set_split_tail(h, 0) turns them off (A/B runs, tests)
    EmArgs ma = a;
    JointExtras mj = jx;
    ma.B = a.B - r;
    ma.T_total = a.T;
    ma.split_groups = G;
    ma.split_window = window;
    ma.b_first = a.B - r;
    ma.xcount = reinterpret_cast<unsigned*>(cfg.xbuf);
    ma.xerror = reinterpret_cast<int*>(cfg.xbuf + 128);
    ma.xslab = reinterpret_cast<double*>(cfg.xbuf + 256);
    mj.main_grid = (int)ma.B;  // <= 2 workgroups per CU: every main problem has its own block
    hipLaunchKernelGGL(kfn, dim3((unsigned)(ma.B + r * G)), dim3(kEmThreads), lds, stream, ma, mj);
    return hipGetLastError() == hipSuccess ? PBBSS_OK : PBBSS_ERR_HIP;
  }
  int64_t grid = (int64_t)cfg.num_cu * occ;
  if (grid > a.B) grid = a.B;
  JointExtras pj = jx;
  pj.main_grid = 0;
  hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(kEmThreads), lds, stream, a, pj);
  return hipGetLastError() == hipSuccess ? PBBSS_OK : PBBSS_ERR_HIP;
}

template <typename YS>
static int launch_joint_k(int K, const EmArgs& a, const JointExtras& jx, int inline_pa,
                          const EmLaunchCfg& cfg, hipStream_t stream) {
#ifdef PBBSS_EM_DEV_ONLY_K  // kernel-development builds: one (K, float) instantiation
  if constexpr (std::is_same<YS, float>::value) {
    if (K == PBBSS_EM_DEV_ONLY_K)
      return launch_joint<PBBSS_EM_DEV_ONLY_K, float>(a, jx, inline_pa, cfg, stream);
  }
  return PBBSS_ERR_UNSUPPORTED;
#else
  switch (K) {
    case 1: return launch_joint<1, YS>(a, jx, inline_pa, cfg, stream);
    case 2: return launch_joint<2, YS>(a, jx, inline_pa, cfg, stream);
    case 3: return launch_joint<3, YS>(a, jx, inline_pa, cfg, stream);
    case 4: return launch_joint<4, YS>(a, jx, inline_pa, cfg, stream);
    case 5: return launch_joint<5, YS>(a, jx, inline_pa, cfg, stream);
    case 6: return launch_joint<6, YS>(a, jx, inline_pa, cfg, stream);
    default: return PBBSS_ERR_UNSUPPORTED;
  }
#endif
}

// Spatial half of the rotated joint loop (run_joint_ms): one workgroup per bin, plain grid.
template <int K, typename YS>
static int launch_joint_ms(const EmArgs& a, const JointMs& jm0, const EmLaunchCfg& cfg,
                           hipStream_t stream) {
  using Kern = EmKernel<PBBSS_EM_D, K, YS, false>;
  size_t lds = Kern::lds_bytes(a.T);
  if (lds > cfg.lds_limit) return PBBSS_ERR_LDS_CAPACITY;
  // the helper blocks of the in-launch spectral finalize carve their totals out of the same
  // dynamic LDS
  const size_t fin_lds = (size_t)(kSpectralFinMaxW2 + 1) * sizeof(double);
  if (jm0.mode != 0 && jm0.fin.kind >= 0 && jm0.fin.helpers > 0 && lds < fin_lds) lds = fin_lds;
  JointMs jm = jm0;
  jm.lds_given = (unsigned)lds;
  auto go = [&](auto kfn) -> int {
    if (!raise_lds_attribute(reinterpret_cast<const void*>(kfn), lds)) return PBBSS_ERR_HIP;
    static thread_local int occ_cache[3] = {0, 0, 0};
    static thread_local size_t occ_lds[3] = {0, 0, 0};
    int& occ = occ_cache[jm.mode];
    if (occ == 0 || occ_lds[jm.mode] != lds) {
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kfn, kEmThreads, lds) != hipSuccess)
        return PBBSS_ERR_HIP;
      if (occ < 1) occ = 1;
      occ_lds[jm.mode] = lds;
    }
    int64_t grid = (int64_t)cfg.num_cu * occ;
    if (grid > a.B) grid = a.B;
    jm.main_grid = (int)grid;
    const int helpers = (jm.mode != 0 && jm.fin.kind >= 0) ? jm.fin.helpers : 0;
    if (helpers > 0 && (lds < (size_t)(kSpectralFinMaxW2 + 1) * sizeof(double) ||
                        2 * jm.fin.K * (jm.fin.E + 1) > kSpectralFinMaxW2))
      return PBBSS_ERR_INTERNAL;  // the caller checks joint_ms_fin_supported first
    hipLaunchKernelGGL(kfn, dim3((unsigned)(grid + helpers)), dim3(kEmThreads), lds, stream, a, jm);
    return hipGetLastError() == hipSuccess ? PBBSS_OK : PBBSS_ERR_HIP;
  };
  switch (jm.mode) {
    case 0: return go(cacgmm_joint_ms_kernel<PBBSS_EM_D, K, YS, 0>);
    case 1: return go(cacgmm_joint_ms_kernel<PBBSS_EM_D, K, YS, 1>);
    case 2: return go(cacgmm_joint_ms_kernel<PBBSS_EM_D, K, YS, 2>);
    default: return PBBSS_ERR_INVALID_ARG;
  }
}

template <typename YS>
static int launch_joint_ms_k(int K, const EmArgs& a, const JointMs& jm, const EmLaunchCfg& cfg,
                             hipStream_t stream) {
#ifdef PBBSS_EM_DEV_ONLY_K
  if constexpr (std::is_same<YS, float>::value) {
    if (K == PBBSS_EM_DEV_ONLY_K) return launch_joint_ms<PBBSS_EM_DEV_ONLY_K, float>(a, jm, cfg, stream);
  }
  return PBBSS_ERR_UNSUPPORTED;
#else
  switch (K) {
    case 1: return launch_joint_ms<1, YS>(a, jm, cfg, stream);
    case 2: return launch_joint_ms<2, YS>(a, jm, cfg, stream);
    case 3: return launch_joint_ms<3, YS>(a, jm, cfg, stream);
    case 4: return launch_joint_ms<4, YS>(a, jm, cfg, stream);
    case 5: return launch_joint_ms<5, YS>(a, jm, cfg, stream);
    case 6: return launch_joint_ms<6, YS>(a, jm, cfg, stream);
    default: return PBBSS_ERR_UNSUPPORTED;
  }
#endif
}

#define PBBSS_CAT2(a, b) a##b
#define PBBSS_CAT(a, b) PBBSS_CAT2(a, b)

int PBBSS_CAT(joint_launch_d, PBBSS_EM_D)(int K, int y_is_c128, const EmArgs& a,
                                          const JointExtras& jx, int inline_pa,
                                          const EmLaunchCfg& cfg, hipStream_t stream) {
  return y_is_c128 ? launch_joint_k<double>(K, a, jx, inline_pa, cfg, stream)
                   : launch_joint_k<float>(K, a, jx, inline_pa, cfg, stream);
}

int PBBSS_CAT(joint_ms_launch_d, PBBSS_EM_D)(int K, int y_is_c128, const EmArgs& a,
                                             const JointMs& jm, const EmLaunchCfg& cfg,
                                             hipStream_t stream) {
  return y_is_c128 ? launch_joint_ms_k<double>(K, a, jm, cfg, stream)
                   : launch_joint_ms_k<float>(K, a, jm, cfg, stream);
}

}  // namespace pbbss
