// One translation unit per sensor count D (compiled with -DPBBSS_EM_D=<D>) so the
// heavy persistent-EM template instantiates in parallel under `make -j`.
#include "cacgmm_em.hpp"
#include "em_launch.hpp"

#ifndef PBBSS_EM_D
#error "compile with -DPBBSS_EM_D=<sensors>"
#endif

namespace pbbss {

// workgroups of the EM kernel per CU at this LDS request (cached per thread: a host round trip)
template <int K, typename YS, bool SPILL>
static int em_occupancy(size_t lds) {
  auto kfn = cacgmm_em_kernel<PBBSS_EM_D, K, YS, SPILL>;
  static thread_local size_t cached_lds = 0;
  static thread_local int cached_occ = 0, cached_dev = -1;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (cached_lds != lds || cached_dev != dev) {
    int q = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, kfn, kEmThreads, lds) != hipSuccess)
      return -1;
    cached_occ = q < 1 ? 1 : q;
    cached_lds = lds;
    cached_dev = dev;
  }
  return cached_occ;
}

template <int K, typename YS, bool SPILL>
static int launch_variant(EmArgs a, const EmLaunchCfg& cfg, hipStream_t stream) {
  using Kern = EmKernel<PBBSS_EM_D, K, YS, SPILL>;
  const size_t lds = Kern::lds_bytes(a.T);
  if (lds > cfg.lds_limit) return PBBSS_ERR_LDS_CAPACITY;
  auto kfn = cacgmm_em_kernel<PBBSS_EM_D, K, YS, SPILL>;
  // the attribute and the occupancy query are host round trips of several microseconds each:
  // not once per launch (the GPU idles meanwhile).  The attribute belongs to the function on a
  // device, i.e. to all threads: it is only ever RAISED (process-wide maximum per device), so a
  // launch of another thread with a smaller request cannot pull it below what this one needs.
  if (!raise_lds_attribute(reinterpret_cast<const void*>(kfn), lds)) return PBBSS_ERR_HIP;
  const int occ = em_occupancy<K, YS, SPILL>(lds);
  if (occ < 1) return PBBSS_ERR_HIP;
  int64_t grid = (int64_t)cfg.num_cu * occ;
  if (grid > a.B) grid = a.B;
  if (SPILL) {
    a.scratch_stride = Kern::scratch_bytes(a.T);
    a.scratch = static_cast<char*>(cfg.get_scratch(cfg.scratch_ctx, a.scratch_stride * grid));
    if (!a.scratch) return PBBSS_ERR_HIP;
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

// Remainder problems [b_first, b_first + r) as split groups on the side stream,
// concurrent with the main launch (fork/join through events).
template <int K, typename YS>
static int launch_split(EmArgs a, int64_t b_first, int r, const EmLaunchCfg& cfg,
                        hipStream_t stream) {
  using Kern = EmKernel<PBBSS_EM_D, K, YS, false>;
  const int window = cfg.split_window > 256 ? 256 : cfg.split_window;  // one E pass per window
  const int G = (a.T + window - 1) / window;
  const size_t lds = Kern::lds_bytes(window);
  const size_t slab_bytes = Kern::split_slab_doubles(r, G) * sizeof(double);
  const size_t head = 256;  // counters (r uint) + error word
  if (head + slab_bytes > cfg.xbuf_bytes) return PBBSS_ERR_UNSUPPORTED;
  auto kfn = cacgmm_em_split_kernel<PBBSS_EM_D, K, YS>;
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
  // fork: everything enqueued on `stream` before the event (the inputs) precedes the side stream
  if (hipStreamWaitEvent(cfg.side_stream, cfg.ev_fork, 0) != hipSuccess) return PBBSS_ERR_HIP;
  // no memsets: the arrival counters are put back to zero by the last member to leave, the error
  // word is stamped with this launch's epoch, member 0 zeroes the status words of its problem
  a.xepoch = next_split_epoch(cfg);
  a.xbuf_given = (unsigned)cfg.xbuf_bytes;
  hipLaunchKernelGGL(kfn, dim3((unsigned)(r * G)), dim3(kEmThreads), lds, cfg.side_stream, a);
  if (hipGetLastError() != hipSuccess) return PBBSS_ERR_HIP;
  if (hipEventRecord(cfg.ev_join, cfg.side_stream) != hipSuccess) return PBBSS_ERR_HIP;
  return PBBSS_OK;
}

template <int K, typename YS>
static int launch_one(const EmArgs& a, const EmLaunchCfg& cfg, hipStream_t stream) {
  if (EmKernel<PBBSS_EM_D, K, YS, false>::lds_bytes(a.T) > cfg.lds_limit)
    return launch_variant<K, YS, true>(a, cfg, stream);  // long utterance: frames in HBM/L2
  // Tail handling: with B = m * num_cu + r (small r) the r extra problems would add a
  // full workgroup to r CUs and set the kernel time; split them over many CUs instead.
  const int64_t r = a.B % cfg.num_cu;
  const int window = cfg.split_window > 256 ? 256 : cfg.split_window;
  const size_t slab_need =
      256 + EmKernel<PBBSS_EM_D, K, YS, false>::split_slab_doubles((int)r, (a.T + window - 1) /
                                                                               window) *
                sizeof(double);
  const bool split = cfg.allow_split && a.iterations >= kSplitMinIterations && a.B > cfg.num_cu &&
                     a.B <= 3 * (int64_t)cfg.num_cu && r >= 1 && r <= kSplitMaxProblems &&
                     a.T >= 2 * cfg.split_window && a.wt == 0 && slab_need <= cfg.xbuf_bytes;
  if (!split) return launch_variant<K, YS, false>(a, cfg, stream);
  EmArgs main_a = a;
  main_a.B = a.B - r;
  // The main launch goes out FIRST: the side-stream preparation of the split groups (fork wait,
  // launch, join record) is a handful of host calls during which the device would otherwise idle;
  // the members are independent of the main workgroups and finish long before them.
  // (hipExtAnyOrderLaunch -- the split kernel as a barrier-less packet of the SAME queue -- would
  // need neither stream nor events, but is documented as unsupported on gfx9.)
  if (hipEventRecord(cfg.ev_fork, stream) != hipSuccess) return PBBSS_ERR_HIP;
  int rc = launch_variant<K, YS, false>(main_a, cfg, stream);
  if (rc != PBBSS_OK) return rc;
  rc = launch_split<K, YS>(a, a.B - r, (int)r, cfg, stream);
  if (rc != PBBSS_OK) return rc;
  if (hipStreamWaitEvent(stream, cfg.ev_join, 0) != hipSuccess) return PBBSS_ERR_HIP;
  return PBBSS_OK;
}

template <typename YS>
static int launch_k(int K, const EmArgs& a, const EmLaunchCfg& cfg, hipStream_t stream) {
#ifdef PBBSS_EM_DEV_ONLY_K  // kernel-development builds: one (K, float) instantiation, 10x faster to compile
  if constexpr (std::is_same<YS, float>::value) {
    if (K == PBBSS_EM_DEV_ONLY_K) return launch_one<PBBSS_EM_DEV_ONLY_K, float>(a, cfg, stream);
  }
  return PBBSS_ERR_UNSUPPORTED;
#else
  switch (K) {
    case 1: return launch_one<1, YS>(a, cfg, stream);
    case 2: return launch_one<2, YS>(a, cfg, stream);
    case 3: return launch_one<3, YS>(a, cfg, stream);
    case 4: return launch_one<4, YS>(a, cfg, stream);
    case 5: return launch_one<5, YS>(a, cfg, stream);
    case 6: return launch_one<6, YS>(a, cfg, stream);
    default: return PBBSS_ERR_UNSUPPORTED;
  }
#endif
}

#define PBBSS_CAT2(a, b) a##b
#define PBBSS_CAT(a, b) PBBSS_CAT2(a, b)

int PBBSS_CAT(em_launch_d, PBBSS_EM_D)(int K, int y_is_c128, const EmArgs& a,
                                       const EmLaunchCfg& cfg, hipStream_t stream) {
  return y_is_c128 ? launch_k<double>(K, a, cfg, stream) : launch_k<float>(K, a, cfg, stream);
}

}  // namespace pbbss
