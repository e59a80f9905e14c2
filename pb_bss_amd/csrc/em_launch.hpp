// Host-side dispatch of the persistent EM kernel over (D, K, storage type).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <atomic>
#include <mutex>
#include <vector>
#include "cacgmm_em.hpp"
#include "cwmm.hpp"

namespace pbbss {

// Split groups for the remainder bins pay a fixed price per LAUNCH (fork / join across streams
// and the members' first hand-offs, ~10 us) against ~5 us per ITERATION that an extra full
// workgroup on one compute unit costs: launches of one or two iterations -- the E- and M-steps of
// the step-wise loop -- run without them (inline aligner loop 387 -> 376 us per iteration).
constexpr int kSplitMinIterations = 3;

struct EmLaunchCfg {
  int num_cu;        // compute units of the bound device
  size_t lds_limit;  // usable LDS bytes per workgroup
  // grow-only device scratch owned by the handle (frame arrays of long utterances)
  void* (*get_scratch)(void* ctx, size_t bytes);
  void* scratch_ctx;
  // split-bin remainder launches: side stream + fork/join events + exchange buffer
  hipStream_t side_stream;
  hipEvent_t ev_fork, ev_join;
  char* xbuf;        // [xbuf_bytes] device memory: counters, error word, slabs
  size_t xbuf_bytes;
  int allow_split;   // 0 disables the split variant (tests / debugging)
  int split_window;  // frames per workgroup of a split problem (multiple of 64)
  int split_prio;    // s_setprio level of the split waves (float64 kernels)
  int split_prio32;  // ... of the packed-FP32 kernel's member workgroups
  int* split_epoch;  // host counter stamping the launches of the split protocol
  unsigned spin_limit;  // EmArgs::spin_limit of every launch (0: the kernels' defaults)
  // kernel timing (pbbss_set_timing): start / stop events attached to the DISPATCH of the EM kernel
  // itself (hipExtLaunchKernelGGL: timestamps of the kernel's own completion signal) instead of
  // two hipEventRecord packets around the call -- those cost ~30 us of queue time per step
  // (tools/launch_gap.py).  Null: timing off.
  hipEvent_t ev_t0, ev_t1;
};

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per function and device, shared by all host
// threads: keep a process-wide monotonic maximum per (function, device) and only ever raise it.
inline bool raise_lds_attribute(const void* fn, size_t lds) {
  constexpr int kMaxDev = 64;
  struct Slot {
    const void* fn;
    std::atomic<size_t> have[kMaxDev];
  };
  static std::mutex mu;
  static std::vector<Slot*> slots;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= kMaxDev)
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
  Slot* sl = nullptr;
  {
    std::lock_guard<std::mutex> g(mu);
    for (Slot* x : slots)
      if (x->fn == fn) sl = x;
    if (!sl) {
      sl = new Slot();
      sl->fn = fn;
      for (auto& h : sl->have) h.store(0);
      slots.push_back(sl);
    }
    if (sl->have[dev].load() >= lds) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return false;
    sl->have[dev].store(lds);
  }
  return true;
}


// launch stamp of the split protocol: never 0 (= "launcher clears the word") nor 1 (what the
// cooperative shared-weight kernel writes)
inline int next_split_epoch(const EmLaunchCfg& cfg) {
  int& e = *cfg.split_epoch;
  e = (e >= 0x7ffffff0 || e < 2) ? 2 : e + 1;
  return e;
}


constexpr int kSplitWindow = 64;      // default frames per workgroup of a split problem
constexpr int kSplitMaxProblems = 8;  // at most this many remainder problems are split

// defined in em_inst.hip, one per compiled D
int em_launch_d2(int K, int y_is_c128, const EmArgs&, const EmLaunchCfg&, hipStream_t);
int em_launch_d3(int K, int y_is_c128, const EmArgs&, const EmLaunchCfg&, hipStream_t);
int em_launch_d4(int K, int y_is_c128, const EmArgs&, const EmLaunchCfg&, hipStream_t);
int em_launch_d5(int K, int y_is_c128, const EmArgs&, const EmLaunchCfg&, hipStream_t);
int em_launch_d6(int K, int y_is_c128, const EmArgs&, const EmLaunchCfg&, hipStream_t);
int em_launch_d7(int K, int y_is_c128, const EmArgs&, const EmLaunchCfg&, hipStream_t);
int em_launch_d8(int K, int y_is_c128, const EmArgs&, const EmLaunchCfg&, hipStream_t);

inline int em_launch(int D, int K, int y_is_c128, const EmArgs& a, const EmLaunchCfg& cfg,
                     hipStream_t s) {
  switch (D) {
    case 2: return em_launch_d2(K, y_is_c128, a, cfg, s);
    case 3: return em_launch_d3(K, y_is_c128, a, cfg, s);
    case 4: return em_launch_d4(K, y_is_c128, a, cfg, s);
    case 5: return em_launch_d5(K, y_is_c128, a, cfg, s);
    case 6: return em_launch_d6(K, y_is_c128, a, cfg, s);
    case 7: return em_launch_d7(K, y_is_c128, a, cfg, s);
    case 8: return em_launch_d8(K, y_is_c128, a, cfg, s);
    default: return PBBSS_ERR_UNSUPPORTED;
  }
}


// packed-FP32 (reference-precision) EM kernels (em32_inst.hip), one per compiled D; complex64 only
int em32_launch_d2(int K, const EmArgs&, const EmLaunchCfg&, hipStream_t);
int em32_launch_d3(int K, const EmArgs&, const EmLaunchCfg&, hipStream_t);
int em32_launch_d4(int K, const EmArgs&, const EmLaunchCfg&, hipStream_t);
int em32_launch_d5(int K, const EmArgs&, const EmLaunchCfg&, hipStream_t);
int em32_launch_d6(int K, const EmArgs&, const EmLaunchCfg&, hipStream_t);
int em32_launch_d7(int K, const EmArgs&, const EmLaunchCfg&, hipStream_t);
int em32_launch_d8(int K, const EmArgs&, const EmLaunchCfg&, hipStream_t);

inline int em32_launch(int D, int K, const EmArgs& a, const EmLaunchCfg& cfg, hipStream_t s) {
  switch (D) {
    case 2: return em32_launch_d2(K, a, cfg, s);
    case 3: return em32_launch_d3(K, a, cfg, s);
    case 4: return em32_launch_d4(K, a, cfg, s);
    case 5: return em32_launch_d5(K, a, cfg, s);
    case 6: return em32_launch_d6(K, a, cfg, s);
    case 7: return em32_launch_d7(K, a, cfg, s);
    case 8: return em32_launch_d8(K, a, cfg, s);
    default: return PBBSS_ERR_UNSUPPORTED;
  }
}

// complex-Watson mixture kernels (cw_inst.hip), one per compiled D
int cw_launch_d2(int K, int y_is_c128, const WatsonArgs&, const EmLaunchCfg&, hipStream_t);
int cw_launch_d3(int K, int y_is_c128, const WatsonArgs&, const EmLaunchCfg&, hipStream_t);
int cw_launch_d4(int K, int y_is_c128, const WatsonArgs&, const EmLaunchCfg&, hipStream_t);
int cw_launch_d5(int K, int y_is_c128, const WatsonArgs&, const EmLaunchCfg&, hipStream_t);
int cw_launch_d6(int K, int y_is_c128, const WatsonArgs&, const EmLaunchCfg&, hipStream_t);
int cw_launch_d7(int K, int y_is_c128, const WatsonArgs&, const EmLaunchCfg&, hipStream_t);
int cw_launch_d8(int K, int y_is_c128, const WatsonArgs&, const EmLaunchCfg&, hipStream_t);

inline int cw_launch(int D, int K, int y_is_c128, const WatsonArgs& a, const EmLaunchCfg& cfg,
                     hipStream_t s) {
  switch (D) {
    case 2: return cw_launch_d2(K, y_is_c128, a, cfg, s);
    case 3: return cw_launch_d3(K, y_is_c128, a, cfg, s);
    case 4: return cw_launch_d4(K, y_is_c128, a, cfg, s);
    case 5: return cw_launch_d5(K, y_is_c128, a, cfg, s);
    case 6: return cw_launch_d6(K, y_is_c128, a, cfg, s);
    case 7: return cw_launch_d7(K, y_is_c128, a, cfg, s);
    case 8: return cw_launch_d8(K, y_is_c128, a, cfg, s);
    default: return PBBSS_ERR_UNSUPPORTED;
  }
}

}  // namespace pbbss

namespace pbbss {
// joint spatial+spectral step of the cACG half (joint_inst.hip), one per compiled D
int joint_launch_d2(int K, int y_is_c128, const EmArgs&, const JointExtras&, int inline_pa, const EmLaunchCfg&, hipStream_t);
int joint_launch_d3(int K, int y_is_c128, const EmArgs&, const JointExtras&, int inline_pa, const EmLaunchCfg&, hipStream_t);
int joint_launch_d4(int K, int y_is_c128, const EmArgs&, const JointExtras&, int inline_pa, const EmLaunchCfg&, hipStream_t);
int joint_launch_d5(int K, int y_is_c128, const EmArgs&, const JointExtras&, int inline_pa, const EmLaunchCfg&, hipStream_t);
int joint_launch_d6(int K, int y_is_c128, const EmArgs&, const JointExtras&, int inline_pa, const EmLaunchCfg&, hipStream_t);
int joint_launch_d7(int K, int y_is_c128, const EmArgs&, const JointExtras&, int inline_pa, const EmLaunchCfg&, hipStream_t);
int joint_launch_d8(int K, int y_is_c128, const EmArgs&, const JointExtras&, int inline_pa, const EmLaunchCfg&, hipStream_t);

inline int joint_launch(int D, int K, int y_is_c128, const EmArgs& a, const JointExtras& jx,
                        int inline_pa, const EmLaunchCfg& cfg, hipStream_t s) {
  switch (D) {
    case 2: return joint_launch_d2(K, y_is_c128, a, jx, inline_pa, cfg, s);
    case 3: return joint_launch_d3(K, y_is_c128, a, jx, inline_pa, cfg, s);
    case 4: return joint_launch_d4(K, y_is_c128, a, jx, inline_pa, cfg, s);
    case 5: return joint_launch_d5(K, y_is_c128, a, jx, inline_pa, cfg, s);
    case 6: return joint_launch_d6(K, y_is_c128, a, jx, inline_pa, cfg, s);
    case 7: return joint_launch_d7(K, y_is_c128, a, jx, inline_pa, cfg, s);
    case 8: return joint_launch_d8(K, y_is_c128, a, jx, inline_pa, cfg, s);
    default: return PBBSS_ERR_UNSUPPORTED;
  }
}
}  // namespace pbbss

namespace pbbss {
// EM with mixture weights shared by groups of problems (shared_inst.hip), one per compiled D
int em_shared_launch_d2(int K, int y_is_c128, const EmArgs&, const EmLaunchCfg&, hipStream_t);
int em_shared_launch_d3(int K, int y_is_c128, const EmArgs&, const EmLaunchCfg&, hipStream_t);
int em_shared_launch_d4(int K, int y_is_c128, const EmArgs&, const EmLaunchCfg&, hipStream_t);
int em_shared_launch_d5(int K, int y_is_c128, const EmArgs&, const EmLaunchCfg&, hipStream_t);
int em_shared_launch_d6(int K, int y_is_c128, const EmArgs&, const EmLaunchCfg&, hipStream_t);
int em_shared_launch_d7(int K, int y_is_c128, const EmArgs&, const EmLaunchCfg&, hipStream_t);
int em_shared_launch_d8(int K, int y_is_c128, const EmArgs&, const EmLaunchCfg&, hipStream_t);

inline int em_shared_launch(int D, int K, int y_is_c128, const EmArgs& a, const EmLaunchCfg& cfg,
                            hipStream_t s) {
  switch (D) {
    case 2: return em_shared_launch_d2(K, y_is_c128, a, cfg, s);
    case 3: return em_shared_launch_d3(K, y_is_c128, a, cfg, s);
    case 4: return em_shared_launch_d4(K, y_is_c128, a, cfg, s);
    case 5: return em_shared_launch_d5(K, y_is_c128, a, cfg, s);
    case 6: return em_shared_launch_d6(K, y_is_c128, a, cfg, s);
    case 7: return em_shared_launch_d7(K, y_is_c128, a, cfg, s);
    case 8: return em_shared_launch_d8(K, y_is_c128, a, cfg, s);
    default: return PBBSS_ERR_UNSUPPORTED;
  }
}
}  // namespace pbbss

namespace pbbss {
// spatial half of the rotated joint loop (joint_inst.hip: run_joint_ms), one per compiled D
int joint_ms_launch_d2(int K, int y_is_c128, const EmArgs&, const JointMs&, const EmLaunchCfg&, hipStream_t);
int joint_ms_launch_d3(int K, int y_is_c128, const EmArgs&, const JointMs&, const EmLaunchCfg&, hipStream_t);
int joint_ms_launch_d4(int K, int y_is_c128, const EmArgs&, const JointMs&, const EmLaunchCfg&, hipStream_t);
int joint_ms_launch_d5(int K, int y_is_c128, const EmArgs&, const JointMs&, const EmLaunchCfg&, hipStream_t);
int joint_ms_launch_d6(int K, int y_is_c128, const EmArgs&, const JointMs&, const EmLaunchCfg&, hipStream_t);
int joint_ms_launch_d7(int K, int y_is_c128, const EmArgs&, const JointMs&, const EmLaunchCfg&, hipStream_t);
int joint_ms_launch_d8(int K, int y_is_c128, const EmArgs&, const JointMs&, const EmLaunchCfg&, hipStream_t);

inline int joint_ms_launch(int D, int K, int y_is_c128, const EmArgs& a, const JointMs& jm,
                           const EmLaunchCfg& cfg, hipStream_t s) {
  switch (D) {
    case 2: return joint_ms_launch_d2(K, y_is_c128, a, jm, cfg, s);
    case 3: return joint_ms_launch_d3(K, y_is_c128, a, jm, cfg, s);
    case 4: return joint_ms_launch_d4(K, y_is_c128, a, jm, cfg, s);
    case 5: return joint_ms_launch_d5(K, y_is_c128, a, jm, cfg, s);
    case 6: return joint_ms_launch_d6(K, y_is_c128, a, jm, cfg, s);
    case 7: return joint_ms_launch_d7(K, y_is_c128, a, jm, cfg, s);
    case 8: return joint_ms_launch_d8(K, y_is_c128, a, jm, cfg, s);
    default: return PBBSS_ERR_UNSUPPORTED;
  }
}
}  // namespace pbbss
