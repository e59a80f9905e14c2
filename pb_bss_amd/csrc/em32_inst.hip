// Packed-FP32 (reference-precision) EM kernel: one translation unit per sensor count D
// (compiled with -DPBBSS_EM_D=<D>), as em_inst.hip.
#include "cacgmm_em32.hpp"
#include "em_launch.hpp"

#ifndef PBBSS_EM_D
#error "compile with -DPBBSS_EM_D=<sensors>"
#endif

namespace pbbss {

template <int K>
static int launch32(EmArgs a, const EmLaunchCfg& cfg, hipStream_t stream) {
  using Kern = EmKernel32<PBBSS_EM_D, K>;
  const size_t lds = Kern::lds_bytes(a.T);
  if (lds > cfg.lds_limit) return PBBSS_ERR_LDS_CAPACITY;  // no HBM-scratch variant of this kernel
  auto kfn = cacgmm_em32_kernel<PBBSS_EM_D, K>;
  if (!raise_lds_attribute(reinterpret_cast<const void*>(kfn), lds)) return PBBSS_ERR_HIP;
  static thread_local size_t cached_lds = 0;
  static thread_local int cached_occ = 0, cached_dev = -1;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (cached_lds != lds || cached_dev != dev) {
    int q = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, kfn, kEmThreads, lds) != hipSuccess)
      return PBBSS_ERR_HIP;
    cached_occ = q < 1 ? 1 : q;
    cached_lds = lds;
    cached_dev = dev;
  }
  const int occ = cached_occ;
  // Remainder problems (B = m * CUs + r, small r: the 513th bin of an utterance) as split groups:
  // G member workgroups per problem, each with a window of frames, in the SAME grid behind the
  // full workgroups -- they need a free occupancy slot next to them.
  const int64_t r = a.B % cfg.num_cu;
  const int window = cfg.split_window > 256 ? 256 : cfg.split_window;
  const int G = (a.T + window - 1) / window;
  const size_t slab_need = 256 + Kern::Base::split_slab_doubles((int)r, G) * sizeof(double);
  const bool split = cfg.allow_split && a.iterations >= kSplitMinIterations && a.B > cfg.num_cu && r >= 1 &&
                     r <= kSplitMaxProblems && a.T >= 2 * cfg.split_window &&
                     slab_need <= cfg.xbuf_bytes &&
                     (a.B - r) <= (int64_t)cfg.num_cu * (occ - 1);
  int64_t grid = (int64_t)cfg.num_cu * occ;
  if (split) {
    a.B -= r;
    if (grid > a.B) grid = a.B;
    a.main_grid = (int)grid;
    grid += r * G;
    a.T_total = a.T;
    a.split_groups = G;
    a.split_window = window;
    a.split_prio = cfg.split_prio32;
    a.b_first = a.B;
    a.xcount = reinterpret_cast<unsigned*>(cfg.xbuf);
    a.xerror = reinterpret_cast<int*>(cfg.xbuf + 128);
    a.xslab = reinterpret_cast<double*>(cfg.xbuf + 256);
    a.xepoch = next_split_epoch(cfg);
  } else if (grid > a.B) {
    grid = a.B;
  }
  a.lds_given = (unsigned)lds;
  if (a.xcount) a.xbuf_given = (unsigned)cfg.xbuf_bytes;
  if (cfg.ev_t0) {
    hipExtLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(kEmThreads), lds, stream, cfg.ev_t0,
                          cfg.ev_t1, 0, a);
  } else {
    hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(kEmThreads), lds, stream, a);
  }
  return hipGetLastError() == hipSuccess ? PBBSS_OK : PBBSS_ERR_HIP;
}

#define PBBSS_CAT2(a, b) a##b
#define PBBSS_CAT(a, b) PBBSS_CAT2(a, b)

int PBBSS_CAT(em32_launch_d, PBBSS_EM_D)(int K, const EmArgs& a, const EmLaunchCfg& cfg,
                                         hipStream_t stream) {
  switch (K) {
    case 1: return launch32<1>(a, cfg, stream);
    case 2: return launch32<2>(a, cfg, stream);
    case 3: return launch32<3>(a, cfg, stream);
    case 4: return launch32<4>(a, cfg, stream);
    case 5: return launch32<5>(a, cfg, stream);
    case 6: return launch32<6>(a, cfg, stream);
    default: return PBBSS_ERR_UNSUPPORTED;
  }
}

}  // namespace pbbss
