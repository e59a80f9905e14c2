// extern "C" boundary of libpbbss_hip.so (see include/pbbss.h).  Argument
// validation, handle state, kernel dispatch; no numerical code lives here.
#include "pbbss.h"
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include "beamform.hpp"
#include "dhtv.hpp"
#include "embed.hpp"
#include "gauss_full.hpp"
#include "generic.hpp"
#include "generic_bf.hpp"
#include "stft.hpp"
#include "em_launch.hpp"
#include "comm.hpp"

#define PBBSS_API extern "C" __attribute__((visibility("default")))

struct pbbss_handle_s {
  int device;
  pbbss::EmLaunchCfg cfg;
  void* scratch;
  size_t scratch_bytes;
  void* work;         // second grow-only slab: workspaces of the multi-kernel mixture loops
  size_t work_bytes;
  void* comm;         // RCCL communicator of pbbss_comm_create (one rank = this process), or null
  int comm_world, comm_rank;
  int split_epoch;    // launch stamp of the split protocol (em_inst.hip: next_split_epoch)
  void* comm_buf;     // pack / gather buffers of pbbss_allgather_masks: owned by the communicator,
  size_t comm_bytes;  // never shared with the work slab (a collective may still be in flight)
  void* team_buf;     // control words + centroid partials of the DHTV team kernel
  size_t team_bytes;
  int dhtv_team;      // workgroups per utterance (0 = default, 1 = one-workgroup kernel)
  int dhtv_probe;     // all-segments-at-once identity probe in front of the plan (pbbss_set_dhtv_probe)
  unsigned long long* prof;
  int timing;
  float last_ms;
  // timed regions record into a ring of event pairs, so that a caller can read the duration of an
  // OLDER launch without draining the queue (pbbss_kernel_ms_lagged)
  static constexpr int kTimingRing = 4;
  hipEvent_t ring0[kTimingRing], ring1[kTimingRing];
  unsigned ring_seq;  // timed regions started so far
  hipEvent_t gate_ev; // completion of this handle's last launch with inter-workgroup waits
  int gate_dev;       // device index of the residency gate this handle takes part in (-1: none)
};

namespace {
inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Grow-only scratch slab.  Growing synchronises the device (hipFree) -- it
// happens at most a few times per process, for utterances too long for LDS.
void* handle_scratch(void* ctx, size_t bytes) {
  pbbss_handle_t h = static_cast<pbbss_handle_t>(ctx);
  if (bytes <= h->scratch_bytes) return h->scratch;
  if (h->scratch) {
    if (hipDeviceSynchronize() != hipSuccess) return nullptr;
    (void)hipFree(h->scratch);
    h->scratch = nullptr;
    h->scratch_bytes = 0;
  }
  void* p = nullptr;
  if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
  h->scratch = p;
  h->scratch_bytes = bytes;
  return p;
}

void* handle_work(pbbss_handle_t h, size_t bytes) {
  if (bytes <= h->work_bytes) return h->work;
  if (h->work) {
    if (hipDeviceSynchronize() != hipSuccess) return nullptr;
    (void)hipFree(h->work);
    h->work = nullptr;
    h->work_bytes = 0;
  }
  void* p = nullptr;
  if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
  h->work = p;
  h->work_bytes = bytes;
  return p;
}

// bump allocator over the work slab (256-byte aligned pieces)
struct WorkCarver {
  char* base;
  size_t off = 0;
  size_t limit;  // bytes the caller asked handle_work for; fits() checks the pieces against it
  explicit WorkCarver(void* b, size_t bytes = ~(size_t)0) : base(static_cast<char*>(b)), limit(bytes) {}
  bool fits() const { return off <= limit; }
  static size_t pad(size_t n) { return (n + 255) & ~(size_t)255; }
  template <typename T>
  T* take(size_t count) {
    T* p = reinterpret_cast<T*>(base + off);
    off += pad(count * sizeof(T));
    return p;
  }
};

inline int copy_d2d(void* dst, const void* src, size_t bytes, hipStream_t s) {
  if (dst == src || bytes == 0) return PBBSS_OK;
  return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s) == hipSuccess ? PBBSS_OK
                                                                                   : PBBSS_ERR_HIP;
}

// The split-bin groups must run CONCURRENTLY with the main EM launch.  HIP maps streams onto
// a handful of hardware queues round-robin; a plain extra stream can land on the queue of the
// caller's stream (observed after RCCL had created its own streams: the two launches then
// serialise, 1.7 -> 2.4 ms).  A stream of a different (highest) priority lives on a separate
// set of queues.
bool make_side_stream(hipStream_t* out) {
  const bool dbg = getenv("PBBSS_DEBUG") != nullptr;
  int least = 0, greatest = 0;
  hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
  if (dbg) fprintf(stderr, "pbbss: priority range rc=%d least=%d greatest=%d\n", (int)e, least, greatest);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    greatest = 0;
  }
  e = hipStreamCreateWithPriority(out, hipStreamNonBlocking, greatest);
  if (dbg) fprintf(stderr, "pbbss: hipStreamCreateWithPriority rc=%d (%s)\n", (int)e, hipGetErrorString(e));
  if (e == hipSuccess) return true;
  (void)hipGetLastError();
  e = hipStreamCreateWithFlags(out, hipStreamNonBlocking);
  if (dbg) fprintf(stderr, "pbbss: hipStreamCreateWithFlags rc=%d (%s)\n", (int)e, hipGetErrorString(e));
  return e == hipSuccess;
}

// Every entry point runs with the handle's device current (a caller holding tensors on several
// GPUs in one process may have another one selected) and restores the caller's selection.
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(pbbss_handle_t h) {
    if (!h) return;
    if (hipGetDevice(&prev) != hipSuccess) {
      (void)hipGetLastError();
      return;
    }
    if (prev != h->device) switched = (hipSetDevice(h->device) == hipSuccess);
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
};

struct TimedRegion {
  pbbss_handle_t h;
  hipStream_t s;
  int slot = 0;
  TimedRegion(pbbss_handle_t h_, hipStream_t s_) : h(h_), s(s_) {
    if (h->timing) {
      slot = (int)(h->ring_seq++ % pbbss_handle_s::kTimingRing);
      (void)hipEventRecord(h->ring0[slot], s);
    }
  }
  ~TimedRegion() {
    if (h->timing) (void)hipEventRecord(h->ring1[slot], s);
  }
};
// ---------------------------------------------------------------------------------------------
// Residency gate (round 4).  Kernels whose workgroups WAIT for each other -- split groups of a
// remainder bin, the cooperative shared-weight kernel, in-grid members, DHTV teams -- need their
// peers on the chip at the same time.  Two such kernels launched concurrently from two handles
// (host threads / streams) of one process can starve each other: each holds compute-unit slots
// while it waits for peers that only fit once the other one lets go (round 3 measured the
// cooperative kernel "not served" in 2-16 % of the fits that ran beside a packed-FP32 fit,
// profiles/r03_i_coop_contention_probe.txt; the bounded waits turn the stall into a repeat, never
// a hang).  The gate removes the situation instead of riding it out: per device, every launch of
// that kind first waits (stream-ordered, hipStreamWaitEvent) for the completion event of the
// previous one -- whichever handle issued it -- and leaves its own completion event behind.  With
// a single handle on the device the gate does nothing at all (stream order already serialises its
// launches); PBBSS_RESIDENCY_GATE=0 switches it off.  The entry points arm it only for calls that
// CAN launch such a kernel (`needed`: may_split() for the fused fits, always for the cooperative
// shared-weight fit and the DHTV solver): plain fits -- no remainder bin, fewer than three
// iterations, generic-size path, the joint models (their members never wait) -- keep their
// multi-stream concurrency beside other handles.  Within one process the gate orders the GATED
// launches exactly since round 5 (their enqueue is serialised, see the constructor).  It does
// not order ungated work: a plain fit, a joint fit or a generic-size fit of another handle can
// still hold compute units while a gated kernel's members are being placed -- for that case,
// and for work of OTHER processes on the device, the bounded waits and the host-side repeats
// remain the safety net.
struct ResidencyGate {
  static constexpr int kMaxDev = 64;
  struct State {
    std::recursive_mutex mu;  // recursive: a gated entry point may create / destroy a handle
    hipEvent_t last = nullptr;        // completion of the most recent gated launch on this device
    pbbss_handle_t owner = nullptr;   // handle whose gate_ev `last` is
    hipStream_t owner_stream = nullptr;
    int handles = 0;                  // live handles on this device
  };
  static State& state(int dev) {
    static State st[kMaxDev];
    return st[dev < 0 || dev >= kMaxDev ? 0 : dev];
  }
  static bool enabled() {
    static const bool on = [] {
      const char* v = getenv("PBBSS_RESIDENCY_GATE");
      return !(v && v[0] == '0');
    }();
    return on;
  }
  pbbss_handle_t h;
  hipStream_t s;
  bool active;
  ResidencyGate(pbbss_handle_t h_, hipStream_t s_, bool needed = true) : h(h_), s(s_), active(false) {
    if (!needed || !h || h->gate_dev < 0 || !enabled()) return;
    State& st = state(h->gate_dev);
    st.mu.lock();
    if (st.handles < 2) {  // nobody to collide with
      st.mu.unlock();
      return;
    }
    // The device mutex stays held until the destructor has recorded this launch's completion
    // event: the host-side ENQUEUE of gated launches is serialised (the device work is not waited
    // for), so a second thread always finds the event of the launch in front of it.  The lock
    // spans the entry point's body: normally microseconds, but a body that has to GROW the
    // handle's workspace (handle_scratch / handle_work: hipDeviceSynchronize + hipFree +
    // hipMalloc, first call at a larger shape only) does so under the lock, and other threads'
    // gated calls wait behind it once.
    // (Until round 5 the lock was dropped in between: two threads entering together both waited
    // for the same older event and then ran side by side -- the residual "not co-resident" case
    // of tests/test_gpu_contention.py, about one full-suite run in ten.)
    active = true;
    if (st.last && !(st.owner == h && st.owner_stream == s))
      (void)hipStreamWaitEvent(s, st.last, 0);
  }
  ~ResidencyGate() {
    if (!active) return;
    State& st = state(h->gate_dev);
    if (hipEventRecord(h->gate_ev, s) == hipSuccess) {
      st.last = h->gate_ev;
      st.owner = h;
      st.owner_stream = s;
    }
    st.mu.unlock();
  }
  static void on_create(pbbss_handle_t h, int dev) {
    h->gate_dev = -1;
    h->gate_ev = nullptr;
    if (dev < 0 || dev >= kMaxDev) return;
    if (hipEventCreateWithFlags(&h->gate_ev, hipEventDisableTiming) != hipSuccess) {
      h->gate_ev = nullptr;
      return;
    }
    h->gate_dev = dev;
    State& st = state(dev);
    // the gate becomes active with the second handle: whatever the first one has in flight was
    // launched without leaving an event behind -- let it drain once (outside the lock: a gated
    // launch of another thread must not wait behind a device-wide synchronisation)
    bool drain;
    {
      std::lock_guard<std::recursive_mutex> g(st.mu);
      drain = ++st.handles == 2;
    }
    if (drain) (void)hipDeviceSynchronize();
  }
  // Can a fused fit of B problems launch workgroups that wait for each other (split groups /
  // in-grid members of the remainder problems: em_inst.hip, em32_inst.hip, cw_inst.hip)?  A
  // superset of the launchers' own conditions, from the arguments alone.
  static bool may_split(pbbss_handle_t h, int64_t B, int D, int iterations) {
    if (!h) return false;
    const int64_t cu = h->cfg.num_cu > 0 ? h->cfg.num_cu : 256;
    return D <= 8 && iterations >= pbbss::kSplitMinIterations && B > cu && B % cu != 0;
  }
  static void on_destroy(pbbss_handle_t h) {
    if (h->gate_dev < 0) return;
    State& st = state(h->gate_dev);
    {
      std::lock_guard<std::recursive_mutex> g(st.mu);
      --st.handles;
      if (st.owner == h) {
        st.last = nullptr;
        st.owner = nullptr;
        st.owner_stream = nullptr;
      }
    }
    if (h->gate_ev) (void)hipEventDestroy(h->gate_ev);
  }
};
}  // namespace

PBBSS_API int pbbss_version(void) { return PBBSS_VERSION; }

PBBSS_API const char* pbbss_error_string(int code) {
  switch (code) {
    case PBBSS_OK: return "ok";
    case PBBSS_ERR_INVALID_ARG: return "invalid argument";
    case PBBSS_ERR_UNSUPPORTED:
      return "shape not covered by the compiled kernels (2 <= D <= 32 sensors, 8 for LCMV; the "
             "class range of every entry point is stated in pbbss.h)";
    case PBBSS_ERR_HIP: return "HIP runtime error";
    case PBBSS_ERR_LDS_CAPACITY:
      return "observation does not fit the LDS-resident EM kernel (too many frames)";
    case PBBSS_ERR_INTERNAL: return "workspace accounting mismatch inside the library (a bug)";
    default: return "unknown error";
  }
}

PBBSS_API int pbbss_create(pbbss_handle_t* out, int device_id) {
  if (!out) return PBBSS_ERR_INVALID_ARG;
  const bool dbg = getenv("PBBSS_DEBUG") != nullptr;
  // bind to device_id for the allocations below, then give the caller its current device back
  // (every other entry point uses DeviceGuard; a lazily created handle must not move the
  // process's current device)
  int prev_device = -1;
  (void)hipGetDevice(&prev_device);
  struct Restore {
    int dev;
    ~Restore() {
      if (dev >= 0) (void)hipSetDevice(dev);
    }
  } restore{prev_device};
  hipError_t e0 = hipSetDevice(device_id);
  if (dbg) fprintf(stderr, "pbbss: hipSetDevice(%d) rc=%d (%s)\n", device_id, (int)e0, hipGetErrorString(e0));
  if (e0 != hipSuccess) return PBBSS_ERR_HIP;
  hipDeviceProp_t prop;
  e0 = hipGetDeviceProperties(&prop, device_id);
  if (dbg) fprintf(stderr, "pbbss: hipGetDeviceProperties rc=%d (%s)\n", (int)e0, hipGetErrorString(e0));
  if (e0 != hipSuccess) return PBBSS_ERR_HIP;
  pbbss_handle_t h = new pbbss_handle_s();
  h->device = device_id;
  h->cfg.num_cu = prop.multiProcessorCount;
  // gfx950: 160 KiB per CU, one workgroup may take all of it
  size_t lds = prop.maxSharedMemoryPerMultiProcessor ? prop.maxSharedMemoryPerMultiProcessor
                                                     : prop.sharedMemPerBlock;
  if (lds < prop.sharedMemPerBlock) lds = prop.sharedMemPerBlock;
  h->cfg.lds_limit = lds;
  h->cfg.get_scratch = handle_scratch;
  h->cfg.scratch_ctx = h;
  h->cfg.allow_split = 1;
  h->cfg.split_window = pbbss::kSplitWindow;
  // wave priority of the remainder bin's member workgroups (s_setprio): 1, above the full
  // workgroups.  At priority 0 the MAIN kernel gets faster (1.43 -> 1.37 ms; with the full
  // workgroups raised to 1 even 1.26 ms, the 512-bin time) but the members then only harvest idle
  // issue slots and need 1.72 ms for their 100 iterations: the step waits for them
  // (profiles/r03_g_member_priority.txt).  The packed-FP32 kernel's members sit in the same grid,
  // where the kernel time shows it directly: 1.05 ms at 1, 1.29 ms at 0.
  h->cfg.split_prio = 1;
  h->cfg.split_prio32 = 1;
  if (const char* p = getenv("PBBSS_SPLIT_PRIO")) h->cfg.split_prio = h->cfg.split_prio32 = atoi(p);
  h->split_epoch = 1;
  h->cfg.split_epoch = &h->split_epoch;
  h->cfg.spin_limit = 0;
  h->cfg.ev_t0 = nullptr;
  h->cfg.ev_t1 = nullptr;
  if (const char* w = getenv("PBBSS_SPLIT_WINDOW")) {
    int v = atoi(w);
    if (v >= 64 && v % 64 == 0) h->cfg.split_window = v;
  }
  h->cfg.xbuf_bytes = (size_t)1 << 20;
  h->cfg.xbuf = nullptr;
  h->cfg.side_stream = nullptr;
  {
    void* xb = nullptr;
    if (hipMalloc(&xb, h->cfg.xbuf_bytes) != hipSuccess ||
        !make_side_stream(&h->cfg.side_stream) ||
        hipEventCreateWithFlags(&h->cfg.ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->cfg.ev_join, hipEventDisableTiming) != hipSuccess) {
      delete h;
      return PBBSS_ERR_HIP;
    }
    h->cfg.xbuf = static_cast<char*>(xb);
    // arrival counters / error words start at zero (the joint launch's members reset their
    // counter themselves; the EM split launch clears its own before every launch)
    if (hipMemset(xb, 0, 256) != hipSuccess) {
      (void)hipFree(xb);
      (void)hipStreamDestroy(h->cfg.side_stream);
      (void)hipEventDestroy(h->cfg.ev_fork);
      (void)hipEventDestroy(h->cfg.ev_join);
      delete h;
      return PBBSS_ERR_HIP;
    }
  }
  h->scratch = nullptr;
  h->scratch_bytes = 0;
  h->work = nullptr;
  h->work_bytes = 0;
  h->team_bytes = (size_t)4 << 20;
  h->team_buf = nullptr;
  if (hipMalloc(&h->team_buf, h->team_bytes) != hipSuccess) {
    h->team_buf = nullptr;  // the one-workgroup kernel needs none
    h->team_bytes = 0;
  }
  h->dhtv_team = 0;
  h->dhtv_probe = 0;
  h->comm = nullptr;
  h->comm_world = 1;
  h->comm_rank = 0;
  h->comm_buf = nullptr;
  h->comm_bytes = 0;
  if (const char* tv = getenv("PBBSS_DHTV_TEAM")) h->dhtv_team = atoi(tv);
  h->prof = nullptr;
  h->timing = 0;
  h->last_ms = 0.f;
  h->ring_seq = 0;
  for (int i = 0; i < pbbss_handle_s::kTimingRing; ++i) {
    if (hipEventCreate(&h->ring0[i]) != hipSuccess || hipEventCreate(&h->ring1[i]) != hipSuccess) {
      delete h;
      return PBBSS_ERR_HIP;
    }
  }
  ResidencyGate::on_create(h, device_id);
  *out = h;
  return PBBSS_OK;
}

PBBSS_API int pbbss_destroy(pbbss_handle_t h) {
  if (!h) return PBBSS_ERR_INVALID_ARG;
  for (int i = 0; i < pbbss_handle_s::kTimingRing; ++i) {
    (void)hipEventDestroy(h->ring0[i]);
    (void)hipEventDestroy(h->ring1[i]);
  }
  if (h->scratch) (void)hipFree(h->scratch);
  if (h->work) (void)hipFree(h->work);
  if (h->comm) (void)pbbss::comm_destroy(h->comm);
  if (h->comm_buf) (void)hipFree(h->comm_buf);
  if (h->team_buf) (void)hipFree(h->team_buf);
  if (h->cfg.xbuf) (void)hipFree(h->cfg.xbuf);
  if (h->cfg.side_stream) (void)hipStreamDestroy(h->cfg.side_stream);
  (void)hipEventDestroy(h->cfg.ev_fork);
  (void)hipEventDestroy(h->cfg.ev_join);
  ResidencyGate::on_destroy(h);
  delete h;
  return PBBSS_OK;
}

// ---------------------------------------------------------------------------
// Multi-GPU: RCCL communicator in the handle + the mask all-gather (comm.hip)
// ---------------------------------------------------------------------------
PBBSS_API int pbbss_comm_unique_id(void* out_id) {
  if (!out_id) return PBBSS_ERR_INVALID_ARG;
  return pbbss::comm_unique_id(out_id);
}

PBBSS_API int pbbss_comm_create(pbbss_handle_t h, const void* unique_id, int world_size, int rank) {
  DeviceGuard device_guard(h);
  if (!h || !unique_id || world_size < 1 || rank < 0 || rank >= world_size)
    return PBBSS_ERR_INVALID_ARG;
  if (h->comm) return PBBSS_ERR_INVALID_ARG;  // one communicator per handle
  void* c = nullptr;
  const int rc = pbbss::comm_create(unique_id, world_size, rank, &c);
  if (rc != PBBSS_OK) return rc;
  h->comm = c;
  h->comm_world = world_size;
  h->comm_rank = rank;
  return PBBSS_OK;
}

PBBSS_API int pbbss_comm_destroy(pbbss_handle_t h) {
  DeviceGuard device_guard(h);
  if (!h) return PBBSS_ERR_INVALID_ARG;
  // collectives enqueued on any stream of this device finish before their buffers go away
  if (h->comm) (void)hipDeviceSynchronize();
  const int rc = pbbss::comm_destroy(h->comm);
  h->comm = nullptr;
  h->comm_world = 1;
  h->comm_rank = 0;
  if (h->comm_buf) (void)hipFree(h->comm_buf);
  h->comm_buf = nullptr;
  h->comm_bytes = 0;
  return rc;
}

PBBSS_API int pbbss_comm_info(pbbss_handle_t h, int* out_world_size, int* out_rank) {
  if (!h || !out_world_size || !out_rank) return PBBSS_ERR_INVALID_ARG;
  if (!h->comm) return PBBSS_ERR_INVALID_ARG;  // pbbss_comm_create first
  return pbbss::comm_query(h->comm, out_world_size, out_rank);
}

PBBSS_API int pbbss_shard_bounds(int64_t total_bins, int world_size, int rank, int64_t* out_start,
                                 int64_t* out_stop) {
  if (total_bins < 0 || world_size < 1 || rank < 0 || rank >= world_size || !out_start || !out_stop)
    return PBBSS_ERR_INVALID_ARG;
  const int64_t base = total_bins / world_size, extra = total_bins % world_size;
  *out_start = rank * base + (rank < extra ? rank : extra);
  *out_stop = *out_start + base + (rank < extra ? 1 : 0);
  return PBBSS_OK;
}

PBBSS_API int pbbss_allgather_unpack(pbbss_handle_t h, const void* gathered, int elem_bytes,
                                     int world_size, int64_t outer, int64_t total_bins,
                                     int64_t inner, void* out, void* stream) {
  DeviceGuard device_guard(h);
  if (!h || !gathered || !out || world_size < 1 || outer < 0 || total_bins < 0 || inner < 0)
    return PBBSS_ERR_INVALID_ARG;
  if (elem_bytes != 4 && elem_bytes != 8) return PBBSS_ERR_UNSUPPORTED;
  return pbbss::launch_allgather_unpack(gathered, elem_bytes, world_size, outer, total_bins, inner,
                                        out, as_stream(stream));
}

PBBSS_API int pbbss_allgather_masks(pbbss_handle_t h, const void* local, int elem_bytes,
                                    int64_t outer, int64_t total_bins, int64_t inner, void* out,
                                    void* stream) {
  DeviceGuard device_guard(h);
  if (!h || !out || outer < 0 || total_bins < 0 || inner < 0) return PBBSS_ERR_INVALID_ARG;
  if (elem_bytes != 4 && elem_bytes != 8) return PBBSS_ERR_UNSUPPORTED;
  if (!h->comm) return PBBSS_ERR_INVALID_ARG;  // pbbss_comm_create first
  const int world = h->comm_world;
  int64_t lo = 0, hi = 0;
  (void)pbbss_shard_bounds(total_bins, world, h->comm_rank, &lo, &hi);
  const int64_t nloc = hi - lo, pad = (total_bins + world - 1) / world;
  if (nloc > 0 && !local) return PBBSS_ERR_INVALID_ARG;
  const size_t block = (size_t)outer * pad * inner * elem_bytes;
  if (block == 0) return PBBSS_OK;
  // The pack / gather buffers belong to the communicator (grow-only, released by
  // pbbss_comm_destroy / pbbss_destroy): the work slab may be re-carved or reallocated by the next
  // library call on another stream while this collective is still in flight.
  const size_t need = WorkCarver::pad(block) + WorkCarver::pad(block * world);
  if (need > h->comm_bytes) {
    if (h->comm_buf) {
      if (hipDeviceSynchronize() != hipSuccess) return PBBSS_ERR_HIP;
      (void)hipFree(h->comm_buf);
      h->comm_buf = nullptr;
      h->comm_bytes = 0;
    }
    if (hipMalloc(&h->comm_buf, need) != hipSuccess) {
      h->comm_buf = nullptr;
      return PBBSS_ERR_HIP;
    }
    h->comm_bytes = need;
  }
  WorkCarver wc(h->comm_buf);
  char* packed = wc.take<char>(block);
  char* gathered = wc.take<char>(block * world);
  hipStream_t s = as_stream(stream);
  int rc = pbbss::launch_allgather_pack(local, elem_bytes, outer, nloc, pad, inner, packed, s);
  if (rc != PBBSS_OK) return rc;
  rc = pbbss::comm_all_gather_bytes(h->comm, packed, gathered, block, s);
  if (rc != PBBSS_OK) return rc;
  return pbbss::launch_allgather_unpack(gathered, elem_bytes, world, outer, total_bins, inner, out,
                                        s);
}

PBBSS_API int pbbss_set_timing(pbbss_handle_t h, int enable) {
  if (!h) return PBBSS_ERR_INVALID_ARG;
  h->timing = enable ? 1 : 0;
  return PBBSS_OK;
}

PBBSS_API int pbbss_set_phase_profile(pbbss_handle_t h, void* dev_counters) {
  if (!h) return PBBSS_ERR_INVALID_ARG;
  h->prof = static_cast<unsigned long long*>(dev_counters);
  return PBBSS_OK;
}

PBBSS_API int pbbss_set_split_tail(pbbss_handle_t h, int enable) {
  if (!h) return PBBSS_ERR_INVALID_ARG;
  h->cfg.allow_split = enable ? 1 : 0;
  return PBBSS_OK;
}

PBBSS_API int pbbss_set_dhtv_team(pbbss_handle_t h, int workgroups_per_utterance) {
  if (!h || workgroups_per_utterance < -pbbss::kDhtvTeamMax || workgroups_per_utterance > 64 ||
      workgroups_per_utterance == -1)
    return PBBSS_ERR_INVALID_ARG;
  h->dhtv_team = workgroups_per_utterance;
  return PBBSS_OK;
}

PBBSS_API int pbbss_set_dhtv_probe(pbbss_handle_t h, int enable) {
  if (!h) return PBBSS_ERR_INVALID_ARG;
  if (enable < 0 || enable > 3) return PBBSS_ERR_INVALID_ARG;
  h->dhtv_probe = enable;
  return PBBSS_OK;
}

PBBSS_API int pbbss_set_spin_limit(pbbss_handle_t h, unsigned polls) {
  DeviceGuard device_guard(h);
  if (!h) return PBBSS_ERR_INVALID_ARG;
  h->cfg.spin_limit = polls;
  return PBBSS_OK;  // (the DHTV team kernels take it as a kernel argument since round 6)
}

PBBSS_API int pbbss_split_error(pbbss_handle_t h, int* out_flag) {
  DeviceGuard device_guard(h);
  if (!h || !out_flag) return PBBSS_ERR_INVALID_ARG;
  int v = 0;
  if (hipMemcpy(&v, h->cfg.xbuf + 192, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess)
    return PBBSS_ERR_HIP;
  *out_flag = v;
  return PBBSS_OK;
}

PBBSS_API int pbbss_split_reset(pbbss_handle_t h) {
  DeviceGuard device_guard(h);
  if (!h || !h->cfg.xbuf) return PBBSS_ERR_INVALID_ARG;
  // every launch of this handle must have left the device: a member still running would see its
  // arrival counter vanish.  Then the counters, the per-launch error word and the sticky flag of
  // pbbss_split_error go back to their creation state (a launch that was aborted half-way -- a
  // device fault, a debug-build trap, a timed-out hand-off -- leaves the counters non-zero, and
  // every later split launch of the handle would pass its barriers early or time out).
  if (hipDeviceSynchronize() != hipSuccess) return PBBSS_ERR_HIP;
  if (hipMemset(h->cfg.xbuf, 0, 256) != hipSuccess) return PBBSS_ERR_HIP;
  return PBBSS_OK;
}

PBBSS_API int pbbss_kernel_ms_lagged(pbbss_handle_t h, int lag, float* out_ms) {
  DeviceGuard device_guard(h);
  if (!h || !out_ms) return PBBSS_ERR_INVALID_ARG;
  if (!h->timing) return PBBSS_ERR_INVALID_ARG;
  if (lag < 0 || lag >= pbbss_handle_s::kTimingRing || (unsigned)lag >= h->ring_seq)
    return PBBSS_ERR_INVALID_ARG;
  const int slot = (int)((h->ring_seq - 1 - (unsigned)lag) % pbbss_handle_s::kTimingRing);
  if (hipEventSynchronize(h->ring1[slot]) != hipSuccess) return PBBSS_ERR_HIP;
  if (hipEventElapsedTime(&h->last_ms, h->ring0[slot], h->ring1[slot]) != hipSuccess)
    return PBBSS_ERR_HIP;
  *out_ms = h->last_ms;
  return PBBSS_OK;
}

PBBSS_API int pbbss_last_kernel_ms(pbbss_handle_t h, float* out_ms) {
  return pbbss_kernel_ms_lagged(h, 0, out_ms);
}

PBBSS_API int pbbss_normalize_observation(pbbss_handle_t h, const void* y, int is_c128,
                                          int64_t B, int T, int D, void* out, void* stream) {
  DeviceGuard device_guard(h);
  if (!h || !y || !out || B <= 0 || T <= 0 || D <= 0) return PBBSS_ERR_INVALID_ARG;
  if (B > 65535) {  // grid.y limit: split the batch
    for (int64_t b0 = 0; b0 < B; b0 += 65535) {
      int64_t nb = (B - b0 < 65535) ? (B - b0) : 65535;
      size_t esz = is_c128 ? 16 : 8;
      int rc = pbbss::launch_normalize((const char*)y + (size_t)b0 * T * D * esz, is_c128, nb, T,
                                       D, (char*)out + (size_t)b0 * T * D * esz,
                                       as_stream(stream));
      if (rc != PBBSS_OK) return rc;
    }
    return PBBSS_OK;
  }
  return pbbss::launch_normalize(y, is_c128, B, T, D, out, as_stream(stream));
}

PBBSS_API int pbbss_cacgmm_fit(pbbss_handle_t h, const void* y, int64_t B, int T, int D, int K,
                               const double* gamma0, const void* in_eigvec,
                               const double* in_eigval, const double* in_weight,
                               const double* saliency, const uint8_t* activity,
                               const pbbss_em_opts* o, void* out_eigvec, double* out_eigval,
                               double* out_weight, int32_t* out_status, double* out_affiliation,
                               double* out_quadratic_form, void* stream) {
  DeviceGuard device_guard(h);
  ResidencyGate residency_gate(h, as_stream(stream),
                               o && ResidencyGate::may_split(h, B, D, o->iterations));
  if (!h || !y || !o || B <= 0 || T <= 0) return PBBSS_ERR_INVALID_ARG;
  if (o->iterations <= 0) return PBBSS_ERR_INVALID_ARG;  // cacgmm.py:200
  const bool has_gamma = gamma0 != nullptr;
  const bool has_model = in_eigvec && in_eigval && in_weight;
  if (has_gamma == has_model) return PBBSS_ERR_INVALID_ARG;  // xor, cacgmm.py:190
  if (!out_eigvec || !out_eigval || !out_weight || !out_status) return PBBSS_ERR_INVALID_ARG;
  if (o->covariance_norm < 0 || o->covariance_norm > 2) return PBBSS_ERR_INVALID_ARG;
  if (o->weight_mode < 0 || o->weight_mode > 1) return PBBSS_ERR_INVALID_ARG;
  if (o->precision != PBBSS_PRECISION_F64 && o->precision != PBBSS_PRECISION_F32)
    return PBBSS_ERR_INVALID_ARG;
  if (o->precision == PBBSS_PRECISION_F32 && (D > 8 || K > 6)) return PBBSS_ERR_UNSUPPORTED;
  if (D > 8 || K > 6) {
    // generic-size path (generic.hip; also more than 6 classes at any D): E-step, covariance + weights, eigendecomposition per
    // iteration, enqueued back to back; the model lives in the caller's output buffers
    if (!pbbss::gen_em_supported(D, K)) return PBBSS_ERR_UNSUPPORTED;
    if (o->layout != PBBSS_LAYOUT_TD) return PBBSS_ERR_UNSUPPORTED;
    hipStream_t s = as_stream(stream);
    const size_t nkt = (size_t)B * K * T;
    const size_t nmat = (size_t)B * K;
    const size_t ninv = pbbss::gen_state_doubles((int64_t)nmat, D);
    const size_t ny = (size_t)B * T * D * (o->y_is_c128 ? 16 : 8);
    const bool transpose = B <= 65535;  // frame-contiguous copy for the E-steps of the loop
    const size_t need = 2 * WorkCarver::pad(nkt * 8) + WorkCarver::pad(nmat * D * D * 16) +
                        WorkCarver::pad(ninv * 8) +
                        2 * WorkCarver::pad(nmat * 8) + WorkCarver::pad(nmat * 4) +
                        WorkCarver::pad((size_t)B * 4) + (transpose ? WorkCarver::pad(ny) : 0);
    void* wmem = handle_work(h, need);
    if (!wmem) return PBBSS_ERR_HIP;
    WorkCarver wc(wmem, need);
    double* aff = wc.take<double>(nkt);
    double* mw = wc.take<double>(nkt);  // M-step weights gamma sal / q / |y|^2 of every frame
    double* cov = wc.take<double>(nmat * D * D * 2);
    double* inv = wc.take<double>(ninv);
    double* inv_logdet = wc.take<double>(nmat);
    int32_t* inv_ok = wc.take<int32_t>(nmat);
    int32_t* zero_bin = wc.take<int32_t>((size_t)B);
    double* csum = wc.take<double>(nmat);
    char* yt = transpose ? wc.take<char>(ny) : nullptr;
    if (!wc.fits()) return PBBSS_ERR_INTERNAL;
    TimedRegion tr(h, s);
    const size_t ysz = o->y_is_c128 ? 16 : 8;
    const size_t ld2 = pbbss::gen_state_doubles(1, D);
    // One chain of launches for the bins [b0, b0 + nb): every array is sliced along the bins
    // (nothing couples them under the per-bin weight modes of this entry point).
    auto chain = [&](int64_t b0, int64_t nb, hipStream_t st) -> int {
      const size_t m0 = (size_t)b0 * K, nm = (size_t)nb * K;
      const char* y_c = static_cast<const char*>(y) + (size_t)b0 * T * D * ysz;
      const double* gamma0_c = gamma0 ? gamma0 + m0 * T : nullptr;
      const double* sal_c = saliency ? saliency + (size_t)b0 * T : nullptr;
      const uint8_t* act_c = activity ? activity + m0 * T : nullptr;
      double* evec_c = static_cast<double*>(out_eigvec) + m0 * D * D * 2;
      double* eval_c = out_eigval + m0 * D;
      double* w_c = out_weight + m0;
      int32_t* st_c = out_status + m0;
      double* aff_c = aff + m0 * T;
      double* mw_c = mw + m0 * T;
      double* cov_c = cov + m0 * D * D * 2;
      double* csum_c = csum + m0;
      int32_t* zero_c = zero_bin + b0;
      int32_t* ok_c = inv_ok + m0;
      const pbbss::GenInverseState state{inv + m0 * ld2, inv_logdet + m0, ok_c};
      const pbbss::GenInverseState state_from_eig{inv + m0 * ld2, inv_logdet + m0, nullptr};
      int rc;
      if (has_model) {
        if ((rc = copy_d2d(evec_c, static_cast<const double*>(in_eigvec) + m0 * D * D * 2,
                           nm * D * D * 16, st)) != PBBSS_OK) return rc;
        if ((rc = copy_d2d(eval_c, in_eigval + m0 * D, nm * D * 8, st)) != PBBSS_OK) return rc;
        if ((rc = copy_d2d(w_c, in_weight + m0, nm * 8, st)) != PBBSS_OK) return rc;
      }
      if (hipMemsetAsync(st_c, 0, nm * sizeof(int32_t), st) != hipSuccess) return PBBSS_ERR_HIP;
      // bins with an all-zero frame (flagged by the E-step / the initial weights) never take
      // the inverse fast path
      if (hipMemsetAsync(zero_c, 0, (size_t)nb * sizeof(int32_t), st) != hipSuccess) return PBBSS_ERR_HIP;
      // E-steps read a (B, D, T) copy of the raw observation (lane = frame is then the contiguous
      // axis); the covariance kernel keeps the caller's (B, T, D) array (its lanes span the channels)
      const void* y_e = y_c;
      int layout_e = PBBSS_LAYOUT_TD;
      if (transpose) {
        char* yt_c = yt + (size_t)b0 * T * D * ysz;
        if ((rc = pbbss::launch_gen_transpose(y_c, o->y_is_c128, nb, T, D, yt_c, st)) != PBBSS_OK) return rc;
        y_e = yt_c;
        layout_e = PBBSS_LAYOUT_DT;
      }
      for (int it = 0; it < o->iterations; ++it) {
        const double* g_src = gamma0_c;
        if (it > 0 || has_model) {
          // from the second iteration on, classes whose inverse was accepted skip (V, lambda)
          rc = pbbss::launch_gen_estep(y_e, o->y_is_c128, layout_e, nb, T, D, K, evec_c, eval_c, w_c,
                                       K, 1, 0, act_c, o->affiliation_eps, aff_c, nullptr, nullptr,
                                       st, it > 0 ? state : state_from_eig, sal_c, mw_c, zero_c,
                                       /*raw_dt=*/1);
          if (rc != PBBSS_OK) return rc;
          g_src = aff_c;
        } else {
          rc = pbbss::launch_gen_init_weights(y_e, o->y_is_c128, layout_e, nb, T, D, K, gamma0_c,
                                              sal_c, mw_c, zero_c, st);
          if (rc != PBBSS_OK) return rc;
        }
        rc = pbbss::launch_gen_mstep_cov(y_c, o->y_is_c128, nb, T, D, K, mw_c, g_src, sal_c,
                                         o->weight_mode, csum_c, cov_c, w_c, st);
        if (rc != PBBSS_OK) return rc;
        const bool last = it + 1 == o->iterations;
        if (!last && !o->force_eig) {
          rc = pbbss::launch_gen_inverse(cov_c, (int64_t)nm, D, o->eigenvalue_floor, state.inv,
                                         state.logdet, ok_c, st, zero_c, K);
          if (rc != PBBSS_OK) return rc;
        } else if (!last) {
          if (hipMemsetAsync(ok_c, 0, nm * sizeof(int32_t), st) != hipSuccess) return PBBSS_ERR_HIP;
        }
        // the eigendecomposition the caller sees comes from the last iteration; before that it
        // only runs for the matrices the inverse test rejected.  Status words are those of the
        // last iteration (earlier ones are overwritten).
        rc = pbbss::launch_gen_heev(cov_c, (int64_t)nm, D, o->covariance_norm, o->eigenvalue_floor,
                                    eval_c, evec_c, st_c, h->cfg.lds_limit, st, last ? nullptr : ok_c);
        if (rc != PBBSS_OK) return rc;
      }
      if (o->final_predict && (out_affiliation || out_quadratic_form)) {
        rc = pbbss::launch_gen_estep(y_e, o->y_is_c128, layout_e, nb, T, D, K, evec_c, eval_c, w_c, K,
                                     1, 0, nullptr, 0.0,
                                     out_affiliation ? out_affiliation + m0 * T : nullptr,
                                     out_quadratic_form ? out_quadratic_form + m0 * T : nullptr,
                                     nullptr, st, state_from_eig, nullptr, nullptr, nullptr,
                                     /*raw_dt=*/1);
        if (rc != PBBSS_OK) return rc;
      }
      return PBBSS_OK;
    };
    // 2^n + 1 bins (every standard STFT size): the last bin turns a whole number of residency
    // rounds into that number plus one straggler in EVERY kernel of the loop (513 bins x 8
    // E-step wavefronts = 4104 on 4096 slots; 513 x 6 covariance workgroups on 768 slots).  A few
    // remainder bins therefore run as their own chain on the side stream, concurrently with the
    // main chain -- forked once, joined once: the bins do not interact inside the loop.
    const int64_t per_round = h->cfg.num_cu > 0 ? h->cfg.num_cu : 256;
    const int64_t rem = B % per_round;
    const bool peel = h->cfg.side_stream && h->cfg.allow_split && B > per_round && rem > 0 &&
                      rem * 16 <= per_round;
    if (!peel) return chain(0, B, s);
    if (hipEventRecord(h->cfg.ev_fork, s) != hipSuccess) return PBBSS_ERR_HIP;
    if (hipStreamWaitEvent(h->cfg.side_stream, h->cfg.ev_fork, 0) != hipSuccess) return PBBSS_ERR_HIP;
    int rc = chain(0, B - rem, s);
    const int rc2 = chain(B - rem, rem, h->cfg.side_stream);
    // join even after an error on one side: the caller's stream must cover everything enqueued
    if (hipEventRecord(h->cfg.ev_join, h->cfg.side_stream) != hipSuccess) return PBBSS_ERR_HIP;
    if (hipStreamWaitEvent(s, h->cfg.ev_join, 0) != hipSuccess) return PBBSS_ERR_HIP;
    return rc != PBBSS_OK ? rc : rc2;
  }
  if (D < 2 || D > 8 || K < 1 || K > 6) return PBBSS_ERR_UNSUPPORTED;
  const bool f32 = o->precision == PBBSS_PRECISION_F32;
  if (f32 && (o->y_is_c128 || out_quadratic_form)) return PBBSS_ERR_UNSUPPORTED;
  pbbss::EmArgs a{};
  a.y = y;
  a.B = B;
  a.T = T;
  a.gamma0 = gamma0;
  a.in_eigvec = static_cast<const double*>(in_eigvec);
  a.in_eigval = in_eigval;
  a.in_weight = in_weight;
  a.wb = K;
  a.wk = 1;
  a.wt = 0;
  a.saliency = saliency;
  a.activity = activity;
  a.out_eigvec = static_cast<double*>(out_eigvec);
  a.out_eigval = out_eigval;
  a.out_weight = out_weight;
  a.out_status = out_status;
  a.out_aff = out_affiliation;
  a.out_q = out_quadratic_form;
  a.iterations = o->iterations;
  a.covariance_norm = o->covariance_norm;
  a.weight_mode = o->weight_mode;
  a.layout = o->layout;
  a.final_predict = o->final_predict && (out_affiliation || out_quadratic_form);
  a.force_eig = o->force_eig;
  a.aff_eps = o->affiliation_eps;
  a.final_eps = 0.0;  // model.predict: affiliation_eps = 0 (cacgmm.py:73)
  a.eig_floor = o->eigenvalue_floor;
  a.prof = h->prof;
  // timing of the fused EM launch: events on the kernel dispatch itself (em_launch.hpp), the
  // duration of the EM kernel as a kernel trace sees it (the split kernel of a remainder bin runs
  // concurrently on the side stream and is shorter)
  pbbss::EmLaunchCfg cfg = h->cfg;
  if (h->timing) {
    const int slot = (int)(h->ring_seq++ % pbbss_handle_s::kTimingRing);
    cfg.ev_t0 = h->ring0[slot];
    cfg.ev_t1 = h->ring1[slot];
  }
  int rc;
  if (f32) {
    rc = pbbss::em32_launch(D, K, a, cfg, as_stream(stream));
    // a long utterance does not fit the LDS-resident packed kernel: say "unsupported", the
    // float64 kernel (which has an HBM-scratch variant) serves it
    if (rc == PBBSS_ERR_LDS_CAPACITY) rc = PBBSS_ERR_UNSUPPORTED;
  } else {
    rc = pbbss::em_launch(D, K, o->y_is_c128, a, cfg, as_stream(stream));
  }
  // a launch that was refused (capacity, shape) recorded no events: give its ring slot back, or
  // pbbss_kernel_ms_lagged would read an unrecorded / stale pair for it
  if (rc != PBBSS_OK && h->timing) --h->ring_seq;
  return rc;
}

PBBSS_API int pbbss_cacgmm_fit_shared(pbbss_handle_t h, const void* y, int64_t B, int T, int D,
                                      int K, int64_t group, const double* gamma0,
                                      const void* in_eigvec, const double* in_eigval,
                                      const double* in_weight, const double* saliency,
                                      const uint8_t* activity, const pbbss_em_opts* o,
                                      void* out_eigvec, double* out_eigval, double* out_weight,
                                      int32_t* out_status, double* out_affiliation,
                                      double* out_quadratic_form, void* stream) {
  DeviceGuard device_guard(h);
  ResidencyGate residency_gate(h, as_stream(stream));  // see ResidencyGate
  if (!h || !y || !o || B <= 0 || T <= 0 || group <= 0 || B % group != 0)
    return PBBSS_ERR_INVALID_ARG;
  if (o->iterations <= 0) return PBBSS_ERR_INVALID_ARG;  // cacgmm.py:200
  const bool has_gamma = gamma0 != nullptr;
  const bool has_model = in_eigvec && in_eigval && in_weight;
  if (has_gamma == has_model) return PBBSS_ERR_INVALID_ARG;  // xor, cacgmm.py:190
  if (!out_eigvec || !out_eigval || !out_weight || !out_status) return PBBSS_ERR_INVALID_ARG;
  if (o->covariance_norm < 0 || o->covariance_norm > 2) return PBBSS_ERR_INVALID_ARG;
  if (o->weight_mode != PBBSS_WEIGHT_SHARED_K && o->weight_mode != PBBSS_WEIGHT_SHARED_KT)
    return PBBSS_ERR_INVALID_ARG;
  if (D < 2 || D > 8 || K < 1 || K > 4 || group > INT32_MAX) return PBBSS_ERR_UNSUPPORTED;
  pbbss::EmArgs a{};
  a.y = y;
  a.B = B;
  a.T = T;
  a.wgroup = (int)group;
  a.gamma0 = gamma0;
  a.in_eigvec = static_cast<const double*>(in_eigvec);
  a.in_eigval = in_eigval;
  a.in_weight = in_weight;
  a.saliency = saliency;
  a.activity = activity;
  a.out_eigvec = static_cast<double*>(out_eigvec);
  a.out_eigval = out_eigval;
  a.out_weight_shared = out_weight;
  a.out_status = out_status;
  a.out_aff = out_affiliation;
  a.out_q = out_quadratic_form;
  a.iterations = o->iterations;
  a.covariance_norm = o->covariance_norm;
  a.weight_mode = o->weight_mode;
  a.layout = o->layout;
  a.final_predict = o->final_predict && (out_affiliation || out_quadratic_form);
  a.force_eig = o->force_eig;
  a.aff_eps = o->affiliation_eps;
  a.final_eps = 0.0;
  a.eig_floor = o->eigenvalue_floor;
  TimedRegion tr(h, as_stream(stream));
  return pbbss::em_shared_launch(D, K, o->y_is_c128, a, h->cfg, as_stream(stream));
}

PBBSS_API int pbbss_cacgmm_predict(pbbss_handle_t h, const void* y, int64_t B, int T, int D,
                                   int K, const void* eigvec, const double* eigval,
                                   const double* weight, int64_t wb, int64_t wk, int64_t wt,
                                   const uint8_t* activity, int layout, int y_is_c128,
                                   double affiliation_eps, double* out_affiliation,
                                   double* out_quadratic_form, double* out_log_pdf,
                                   void* stream) {
  DeviceGuard device_guard(h);
  if (!h || !y || !eigvec || !eigval || !weight || B <= 0 || T <= 0) return PBBSS_ERR_INVALID_ARG;
  if (!out_affiliation && !out_quadratic_form && !out_log_pdf) return PBBSS_ERR_INVALID_ARG;
  if (D > 8 || K > 6) {
    if (!pbbss::gen_em_supported(D, K)) return PBBSS_ERR_UNSUPPORTED;
    const size_t ninv = pbbss::gen_state_doubles(B * K, D);
    void* wmem = handle_work(h, WorkCarver::pad(ninv * 8) + WorkCarver::pad((size_t)B * K * 8));
    if (!wmem) return PBBSS_ERR_HIP;
    WorkCarver wc(wmem);
    double* inv = wc.take<double>(ninv);
    double* inv_logdet = wc.take<double>((size_t)B * K);
    TimedRegion tr(h, as_stream(stream));
    return pbbss::launch_gen_estep(y, y_is_c128, layout, B, T, D, K,
                                   static_cast<const double*>(eigvec), eigval, weight, wb, wk, wt,
                                   activity, affiliation_eps, out_affiliation, out_quadratic_form,
                                   out_log_pdf, as_stream(stream),
                                   pbbss::GenInverseState{inv, inv_logdet, nullptr});
  }
  if (D < 2 || D > 8 || K < 1 || K > 6) return PBBSS_ERR_UNSUPPORTED;
  pbbss::EmArgs a{};
  a.y = y;
  a.B = B;
  a.T = T;
  a.in_eigvec = static_cast<const double*>(eigvec);
  a.in_eigval = eigval;
  a.in_weight = weight;
  a.wb = wb;
  a.wk = wk;
  a.wt = wt;
  a.final_activity = activity;  // CACGMM.predict(source_activity_mask=...)
  a.out_aff = out_affiliation;
  a.out_q = out_quadratic_form;
  a.out_logpdf = out_log_pdf;
  a.iterations = 0;
  a.layout = layout;
  a.final_predict = 1;
  a.final_eps = affiliation_eps;
  a.covariance_norm = PBBSS_COVNORM_EIGENVALUE;
  TimedRegion tr(h, as_stream(stream));
  return pbbss::em_launch(D, K, y_is_c128, a, h->cfg, as_stream(stream));
}

PBBSS_API int pbbss_cacg_m_step(pbbss_handle_t h, const void* y, int64_t B, int T, int D, int K,
                                const double* saliency, const double* quadratic_form, int layout,
                                int y_is_c128, int covariance_norm, double eigenvalue_floor,
                                void* out_eigvec, double* out_eigval, void* out_cov,
                                int32_t* out_status, void* stream) {
  DeviceGuard device_guard(h);
  if (!h || !y || !saliency || B <= 0 || T <= 0) return PBBSS_ERR_INVALID_ARG;
  if (!out_eigvec || !out_eigval || !out_status) return PBBSS_ERR_INVALID_ARG;
  if (covariance_norm < 0 || covariance_norm > 2) return PBBSS_ERR_INVALID_ARG;
  if (D > 8 || K > 6) {
    hipStream_t s = as_stream(stream);
    double* cov = static_cast<double*>(out_cov);
    if (!cov) {
      void* wmem = handle_work(h, (size_t)B * K * D * D * 16);
      if (!wmem) return PBBSS_ERR_HIP;
      cov = static_cast<double*>(wmem);
    }
    int rc = pbbss::launch_gen_cov(y, y_is_c128, layout, B, T, D, K, saliency, (int64_t)K * T,
                                   quadratic_form, nullptr, 0, PBBSS_WEIGHT_PER_CLASS_MEAN, cov,
                                   nullptr, nullptr, h->cfg.lds_limit, s);
    if (rc != PBBSS_OK) return rc;
    return pbbss::launch_gen_heev(cov, B * K, D, covariance_norm, eigenvalue_floor, out_eigval,
                                  static_cast<double*>(out_eigvec), out_status, h->cfg.lds_limit,
                                  s);
  }
  if (D < 2 || D > 8 || K < 1 || K > 6) return PBBSS_ERR_UNSUPPORTED;
  pbbss::EmArgs a{};
  a.y = y;
  a.B = B;
  a.T = T;
  a.gamma0 = saliency;         // the "masked affiliation" (cacgmm.py:332-338)
  a.q0 = quadratic_form;       // null = ones
  a.out_eigvec = static_cast<double*>(out_eigvec);
  a.out_eigval = out_eigval;
  a.out_status = out_status;
  a.out_cov = static_cast<double*>(out_cov);
  a.iterations = 1;
  a.covariance_norm = covariance_norm;
  a.weight_mode = PBBSS_WEIGHT_PER_CLASS_MEAN;
  a.layout = layout;
  a.eig_floor = eigenvalue_floor;
  return pbbss::em_launch(D, K, y_is_c128, a, h->cfg, as_stream(stream));
}

PBBSS_API int pbbss_heev_batched(pbbss_handle_t h, const void* a, int64_t N, int D,
                                 double* out_eigval, void* out_eigvec, int32_t* out_status,
                                 void* stream) {
  DeviceGuard device_guard(h);
  if (!h || !a || !out_eigval || !out_eigvec || N <= 0) return PBBSS_ERR_INVALID_ARG;
  if (D > 8)
    return pbbss::launch_gen_heev(static_cast<const double*>(a), N, D, -1, 0.0, out_eigval,
                                  static_cast<double*>(out_eigvec), out_status, h->cfg.lds_limit,
                                  as_stream(stream));
  return pbbss::launch_heev(static_cast<const double*>(a), N, D, out_eigval,
                            static_cast<double*>(out_eigvec), out_status, as_stream(stream));
}

PBBSS_API int pbbss_psd(pbbss_handle_t h, const void* x, int x_is_c128, int64_t B, int T, int D,
                        int K, const double* mask, int normalize, void* out, void* stream) {
  DeviceGuard device_guard(h);
  if (!h || !x || !out || B <= 0 || T <= 0 || K <= 0) return PBBSS_ERR_INVALID_ARG;
  if (!mask && K != 1) return PBBSS_ERR_INVALID_ARG;
  if (D > 8 || K > 6) {
    return pbbss::launch_gen_cov(x, x_is_c128, PBBSS_LAYOUT_DT, B, T, D, K, mask, (int64_t)K * T,
                                 nullptr, nullptr, (mask && normalize) ? 1 : 2,
                                 PBBSS_WEIGHT_PER_CLASS_MEAN, static_cast<double*>(out), nullptr,
                                 nullptr, h->cfg.lds_limit, as_stream(stream));
  }
  return pbbss::launch_psd(x, x_is_c128, B, T, D, K, mask, normalize, static_cast<double*>(out),
                           h->cfg, as_stream(stream));
}

PBBSS_API int pbbss_gev(pbbss_handle_t h, const void* target, const void* noise, int64_t N,
                        int D, void* out_w, int32_t* out_status, void* stream) {
  DeviceGuard device_guard(h);
  if (!h || !target || !noise || !out_w || N <= 0) return PBBSS_ERR_INVALID_ARG;
  if (D > 8)
    return pbbss::launch_gen_gev(static_cast<const double*>(target),
                                 static_cast<const double*>(noise), N, D,
                                 static_cast<double*>(out_w), out_status, h->cfg.lds_limit,
                                 as_stream(stream));
  return pbbss::launch_gev(static_cast<const double*>(target), static_cast<const double*>(noise),
                           N, D, static_cast<double*>(out_w), out_status, as_stream(stream));
}

PBBSS_API int pbbss_gev_general(pbbss_handle_t h, const void* target, const void* noise,
                                int64_t N, int D, void* out_w, void* out_lambda,
                                int32_t* out_status, void* stream) {
  DeviceGuard device_guard(h);
  if (!h || !target || !noise || !out_w || N <= 0) return PBBSS_ERR_INVALID_ARG;
  return pbbss::launch_gev_general(static_cast<const double*>(target),
                                   static_cast<const double*>(noise), N, D,
                                   static_cast<double*>(out_w), static_cast<double*>(out_lambda),
                                   out_status, as_stream(stream));
}

PBBSS_API int pbbss_solve(pbbss_handle_t h, const void* A, const void* Bm, int64_t N, int D,
                          int M, void* out_x, int32_t* out_status, void* stream) {
  DeviceGuard device_guard(h);
  if (!h || !A || !Bm || !out_x || N <= 0 || M <= 0) return PBBSS_ERR_INVALID_ARG;
  if (D > 8)
    return pbbss::launch_gen_solve(static_cast<const double*>(A), static_cast<const double*>(Bm),
                                   N, D, M, static_cast<double*>(out_x), out_status,
                                   h->cfg.lds_limit, as_stream(stream));
  if (M > 8) return PBBSS_ERR_UNSUPPORTED;
  return pbbss::launch_solve(static_cast<const double*>(A), static_cast<const double*>(Bm), N, D,
                             M, static_cast<double*>(out_x), out_status, as_stream(stream));
}

PBBSS_API int pbbss_mvdr_souden(pbbss_handle_t h, const void* target, const void* noise,
                                int64_t N, int D, double eps, void* out_mat, void* out_snr_num,
                                void* out_snr_den, int32_t* out_status, void* stream) {
  DeviceGuard device_guard(h);
  if (!h || !target || !noise || !out_mat || N <= 0) return PBBSS_ERR_INVALID_ARG;
  if (D > 8)
    return pbbss::launch_gen_souden(static_cast<const double*>(target),
                                    static_cast<const double*>(noise), N, D, eps, 0,
                                    static_cast<double*>(out_mat),
                                    static_cast<double*>(out_snr_num),
                                    static_cast<double*>(out_snr_den), out_status,
                                    h->cfg.lds_limit, as_stream(stream));
  return pbbss::launch_mvdr_souden(static_cast<const double*>(target),
                                   static_cast<const double*>(noise), N, D, eps, 0,
                                   static_cast<double*>(out_mat),
                                   static_cast<double*>(out_snr_num),
                                   static_cast<double*>(out_snr_den), out_status,
                                   as_stream(stream));
}

PBBSS_API int pbbss_mvdr(pbbss_handle_t h, const void* atf, const void* noise, int64_t N, int D,
                         void* out_w, int32_t* out_status, void* stream) {
  DeviceGuard device_guard(h);
  if (!h || !atf || !noise || !out_w || N <= 0) return PBBSS_ERR_INVALID_ARG;
  if (D > 8)
    return pbbss::launch_gen_mvdr(static_cast<const double*>(atf),
                                  static_cast<const double*>(noise), N, D,
                                  static_cast<double*>(out_w), out_status, h->cfg.lds_limit,
                                  as_stream(stream));
  return pbbss::launch_mvdr(static_cast<const double*>(atf), static_cast<const double*>(noise), N,
                            D, static_cast<double*>(out_w), out_status, as_stream(stream));
}

PBBSS_API int pbbss_ban(pbbss_handle_t h, const void* w, const void* noise, int64_t N, int D,
                        void* out_w, void* stream) {
  DeviceGuard device_guard(h);
  if (!h || !w || !noise || !out_w || N <= 0) return PBBSS_ERR_INVALID_ARG;
  if (D > 8)
    return pbbss::launch_gen_ban(static_cast<const double*>(w), static_cast<const double*>(noise),
                                 N, D, static_cast<double*>(out_w), as_stream(stream));
  return pbbss::launch_ban(static_cast<const double*>(w), static_cast<const double*>(noise), N, D,
                           static_cast<double*>(out_w), as_stream(stream));
}

PBBSS_API int pbbss_apply_beamforming_vector(pbbss_handle_t h, const void* w, const void* x,
                                             int x_is_c128, int64_t B, int T, int D, void* out,
                                             void* stream) {
  DeviceGuard device_guard(h);
  if (!h || !w || !x || !out || B <= 0 || T <= 0 || D <= 0) return PBBSS_ERR_INVALID_ARG;
  if (D >= 30) return PBBSS_ERR_INVALID_ARG;  // beamformer.py:582
  for (int64_t b0 = 0; b0 < B; b0 += 65535) {
    int64_t nb = (B - b0 < 65535) ? (B - b0) : 65535;
    size_t esz = x_is_c128 ? 16 : 8;
    int rc = pbbss::launch_apply(static_cast<const double*>(w) + b0 * D * 2,
                                 (const char*)x + (size_t)b0 * D * T * esz, x_is_c128, nb, T, D,
                                 static_cast<double*>(out) + b0 * T * 2, as_stream(stream));
    if (rc != PBBSS_OK) return rc;
  }
  return PBBSS_OK;
}

PBBSS_API int pbbss_apply_beamforming_vector_shared(pbbss_handle_t h, const void* w, const void* x,
                                                    int x_is_c128, int64_t B, int64_t x_batch, int T,
                                                    int D, void* out, void* stream) {
  DeviceGuard device_guard(h);
  if (!h || !w || !x || !out || B <= 0 || T <= 0 || D <= 0 || x_batch <= 0 || B % x_batch != 0)
    return PBBSS_ERR_INVALID_ARG;
  if (D >= 30) return PBBSS_ERR_INVALID_ARG;  // beamformer.py:582
  for (int64_t b0 = 0; b0 < B; b0 += 65535) {
    int64_t nb = (B - b0 < 65535) ? (B - b0) : 65535;
    int rc = pbbss::launch_apply(static_cast<const double*>(w) + b0 * D * 2, x, x_is_c128, nb, T, D,
                                 static_cast<double*>(out) + b0 * T * 2, as_stream(stream), x_batch,
                                 b0);
    if (rc != PBBSS_OK) return rc;
  }
  return PBBSS_OK;
}

PBBSS_API int pbbss_select_reference_channel(pbbss_handle_t h, const void* mat, const void* snr_num,
                                             const void* snr_den, int64_t L, int64_t F, int D,
                                             int64_t lead_stride, int64_t bin_stride, double eps,
                                             void* out_w, int32_t* out_ref, int32_t* out_ok,
                                             void* stream) {
  DeviceGuard device_guard(h);
  if (!h || !mat || !snr_num || !snr_den || !out_w || !out_ref || !out_ok || L <= 0 || F <= 0 ||
      D < 1 || lead_stride < 0 || bin_stride < 0 || out_w == mat)
    return PBBSS_ERR_INVALID_ARG;
  if (D > 32) return PBBSS_ERR_UNSUPPORTED;
  return pbbss::launch_select_reference_channel(
      static_cast<const double*>(mat), static_cast<const double*>(snr_num),
      static_cast<const double*>(snr_den), L, F, D, lead_stride, bin_stride, eps,
      static_cast<double*>(out_w), out_ref, out_ok, as_stream(stream));
}

PBBSS_API int pbbss_dhtv_calculate_mapping(pbbss_handle_t h, const double* mask, int64_t U, int K,
                                           int F, int T, const int32_t* plan, int P, int optimal,
                                           int metric, double* scratch, int32_t* out_mapping,
                                           int32_t* out_status, void* stream) {
  DeviceGuard device_guard(h);
  ResidencyGate residency_gate(h, as_stream(stream));  // see ResidencyGate
  if (!h || !mask || !plan || !scratch || !out_mapping || !out_status) return PBBSS_ERR_INVALID_ARG;
  if (U <= 0 || F <= 0 || T <= 0 || P <= 0) return PBBSS_ERR_INVALID_ARG;
  return pbbss::launch_dhtv(mask, U, K, F, T, plan, P, optimal, metric, scratch, out_mapping, out_status,
                            h->cfg.lds_limit, h->cfg.num_cu, h->dhtv_team, h->team_buf,
                            h->team_bytes, h->dhtv_probe, h->cfg.spin_limit, as_stream(stream));
}

PBBSS_API int pbbss_pa_pairwise_mapping(pbbss_handle_t h, const double* mask,
                                        const double* reference, int64_t U, int K, int64_t F,
                                        int T, const int64_t* mask_strides,
                                        const int64_t* reference_strides, int metric, int optimal,
                                        double* out_scores, int32_t* out_mapping, int64_t map_F,
                                        int64_t map_col0, int32_t* out_status, void* stream) {
  DeviceGuard device_guard(h);
  if (!h || !mask || !reference || !mask_strides || !reference_strides || !out_mapping ||
      !out_status)
    return PBBSS_ERR_INVALID_ARG;
  if (U <= 0 || F <= 0 || T <= 0 || map_col0 < 0 || map_col0 + F > map_F)
    return PBBSS_ERR_INVALID_ARG;
  return pbbss::launch_pa_pair(mask, reference, U, K, F, T, mask_strides, reference_strides,
                               metric, optimal, out_scores, out_mapping, map_F, map_col0,
                               out_status, as_stream(stream));
}

PBBSS_API int pbbss_pa_compose_mapping(pbbss_handle_t h, int32_t* mapping, int64_t U, int K,
                                       int64_t F, void* stream) {
  DeviceGuard device_guard(h);
  if (!h || !mapping || U <= 0 || F <= 0) return PBBSS_ERR_INVALID_ARG;
  return pbbss::launch_pa_compose(mapping, U, K, F, as_stream(stream));
}

PBBSS_API int pbbss_pa_mapping_from_scores(pbbss_handle_t h, const double* scores, int64_t N,
                                           int K, int optimal, int32_t* out_mapping,
                                           int32_t* out_status, void* stream) {
  DeviceGuard device_guard(h);
  if (!h || !scores || !out_mapping || !out_status || N <= 0) return PBBSS_ERR_INVALID_ARG;
  return pbbss::launch_pa_assign(scores, N, K, optimal, out_mapping, out_status,
                                 as_stream(stream));
}

PBBSS_API int pbbss_apply_mapping(pbbss_handle_t h, const double* mask, const int32_t* mapping,
                                  int64_t U, int K, int F, int T, double* out, void* stream) {
  DeviceGuard device_guard(h);
  if (!h || !mask || !mapping || !out || U <= 0 || K <= 0 || F <= 0 || T <= 0)
    return PBBSS_ERR_INVALID_ARG;
  return pbbss::launch_apply_mapping(mask, mapping, U, K, F, T, out, as_stream(stream));
}

PBBSS_API int pbbss_cwmm_fit(pbbss_handle_t h, const void* y, int64_t B, int T, int D, int K,
                             const double* gamma0, const void* in_mode,
                             const double* in_concentration, const double* in_weight,
                             const double* saliency, const pbbss_cwmm_opts* o,
                             const double* spline_t, const double* spline_c, void* out_mode,
                             double* out_concentration, double* out_weight, int32_t* out_status,
                             double* out_affiliation, double* out_log_pdf, void* stream) {
  DeviceGuard device_guard(h);
  // shared class weights run as ONE cooperative launch per batch of groups (cw_launch_shared)
  ResidencyGate residency_gate(
      h, as_stream(stream),
      o && (o->weight_mode == PBBSS_WEIGHT_SHARED_K || o->weight_mode == PBBSS_WEIGHT_SHARED_KT ||
            ResidencyGate::may_split(h, B, D, o->iterations)));
  if (!h || !y || !o || B <= 0 || T <= 0) return PBBSS_ERR_INVALID_ARG;
  if (o->iterations < 0) return PBBSS_ERR_INVALID_ARG;
  const bool has_gamma = gamma0 != nullptr;
  const bool has_model = in_mode && in_concentration && in_weight;
  if (has_gamma == has_model) return PBBSS_ERR_INVALID_ARG;
  if (o->iterations == 0 && !has_model) return PBBSS_ERR_INVALID_ARG;
  if (o->iterations > 0 && (!spline_t || !spline_c || o->n_coef < 3)) return PBBSS_ERR_INVALID_ARG;
  if (o->iterations > 0 && (!out_mode || !out_concentration || !out_weight))
    return PBBSS_ERR_INVALID_ARG;
  // PBBSS_WEIGHT_SHARED_K (round 4): weights averaged over groups of `opts->group` consecutive
  // problems (the bins of an utterance), fused kernels only, fit from affiliations only
  const bool shared_k = o->weight_mode == PBBSS_WEIGHT_SHARED_K ||
                        o->weight_mode == PBBSS_WEIGHT_SHARED_KT;  // (group, K) / (group, K, T)
  if (o->weight_mode < 0 || (o->weight_mode > 1 && !shared_k)) return PBBSS_ERR_INVALID_ARG;
  if (shared_k && (o->group < 1 || B % o->group != 0 || !has_gamma || o->iterations < 1))
    return PBBSS_ERR_INVALID_ARG;
  if (shared_k && (D > 8 || K > 4)) return PBBSS_ERR_UNSUPPORTED;
  if (D > 8 || K > 4) {
    // generic-size path (generic_watson.hip): per iteration class log-pdfs -> softmax with the
    // weights -> masked covariance of the unit-norm frames + weights -> eigh -> principal pair,
    // concentration, ln c; enqueued back to back, the model lives in the work area
    if (!pbbss::gen_supported(D, K) || B > 65535) return PBBSS_ERR_UNSUPPORTED;
    hipStream_t s = as_stream(stream);
    const size_t nkt = (size_t)B * K * T, nmat = (size_t)B * K;
    const size_t need = 2 * WorkCarver::pad(nkt * 8) + 2 * WorkCarver::pad(nmat * D * D * 16) +
                        WorkCarver::pad(nmat * D * 8) + WorkCarver::pad(nmat * D * 16) +
                        3 * WorkCarver::pad(nmat * 8) + WorkCarver::pad(nmat * 4);
    void* wmem = handle_work(h, need);
    if (!wmem) return PBBSS_ERR_HIP;
    WorkCarver wc(wmem, need);
    double* aff = wc.take<double>(nkt);
    double* lp = wc.take<double>(nkt);
    double* cov = wc.take<double>(nmat * D * D * 2);
    double* evec = wc.take<double>(nmat * D * D * 2);
    double* eval = wc.take<double>(nmat * D);
    double* mode_w = wc.take<double>(nmat * D * 2);
    double* conc_w = wc.take<double>(nmat);
    double* weight_w = wc.take<double>(nmat);
    double* lognorm = wc.take<double>(nmat);
    int32_t* status_w = wc.take<int32_t>(nmat);
    if (!wc.fits()) return PBBSS_ERR_INTERNAL;
    double* mode = out_mode ? static_cast<double*>(out_mode) : mode_w;
    double* conc = out_concentration ? out_concentration : conc_w;
    double* weight = out_weight ? out_weight : weight_w;
    int32_t* status = out_status ? out_status : status_w;
    const pbbss::GenWatsonSpline sp{spline_t, spline_c, o->n_coef, o->ev_min, o->ev_max,
                                    o->max_concentration};
    TimedRegion tr(h, s);
    int rc;
    if (hipMemsetAsync(status, 0, nmat * sizeof(int32_t), s) != hipSuccess) return PBBSS_ERR_HIP;
    if (has_model) {
      if ((rc = copy_d2d(mode, in_mode, nmat * D * 16, s)) != PBBSS_OK) return rc;
      if ((rc = copy_d2d(conc, in_concentration, nmat * 8, s)) != PBBSS_OK) return rc;
      if ((rc = copy_d2d(weight, in_weight, nmat * 8, s)) != PBBSS_OK) return rc;
      if ((rc = pbbss::launch_gen_watson_lognorm(conc, (int64_t)nmat, D, lognorm, s)) != PBBSS_OK)
        return rc;
    }
    auto e_step = [&](double* out_lp, double* out_aff) {
      int r = pbbss::launch_gen_watson_logpdf(y, o->y_is_c128, B, T, D, K, mode, conc, lognorm,
                                              out_lp, s);
      if (r != PBBSS_OK || !out_aff) return r;
      return pbbss::launch_log_pdf_to_affiliation(out_lp, B, K, T, weight, K, 1, 0, nullptr, 0.0,
                                                  out_aff, s);
    };
    for (int it = 0; it < o->iterations; ++it) {
      const double* g_src = gamma0;
      if (it > 0 || has_model) {
        if ((rc = e_step(lp, aff)) != PBBSS_OK) return rc;
        g_src = aff;
      }
      rc = pbbss::launch_gen_cov(y, o->y_is_c128, PBBSS_LAYOUT_TD, B, T, D, K, g_src,
                                 (int64_t)K * T, nullptr, saliency, /*mode=*/3, o->weight_mode, cov,
                                 weight, nullptr, h->cfg.lds_limit, s);
      if (rc != PBBSS_OK) return rc;
      rc = pbbss::launch_gen_heev(cov, (int64_t)nmat, D, -1, 0.0, eval, evec, status,
                                  h->cfg.lds_limit, s);
      if (rc != PBBSS_OK) return rc;
      rc = pbbss::launch_gen_watson_finish(eval, evec, (int64_t)nmat, D, sp, mode, conc, lognorm, s);
      if (rc != PBBSS_OK) return rc;
    }
    if (o->final_predict && (out_affiliation || out_log_pdf))
      return e_step(out_log_pdf ? out_log_pdf : lp, out_affiliation);
    return PBBSS_OK;
  }
  if (D < 2 || D > 8 || K < 1 || K > 4) return PBBSS_ERR_UNSUPPORTED;
  pbbss::WatsonArgs wa{};
  wa.em.y = y;
  wa.em.B = B;
  wa.em.T = T;
  wa.em.gamma0 = gamma0;
  wa.em.in_weight = in_weight;
  wa.em.wb = K;
  wa.em.wk = 1;
  wa.em.wt = 0;
  wa.em.saliency = saliency;
  wa.em.out_weight = out_weight;
  wa.em.out_status = out_status;
  wa.em.out_aff = out_affiliation;
  wa.em.out_logpdf = out_log_pdf;
  wa.em.iterations = o->iterations;
  wa.em.weight_mode = o->weight_mode;
  if (shared_k) {  // out_weight is (B / group, K)
    wa.em.wgroup = o->group;
    wa.em.out_weight = nullptr;
    wa.em.out_weight_shared = out_weight;
  }
  wa.em.layout = PBBSS_LAYOUT_TD;
  wa.em.final_predict = o->final_predict && (out_affiliation || out_log_pdf);
  wa.in_mode = static_cast<const double*>(in_mode);
  wa.in_conc = in_concentration;
  wa.spline_t = spline_t;
  wa.spline_c = spline_c;
  wa.n_coef = o->n_coef;
  wa.ev_min = o->ev_min;
  wa.ev_max = o->ev_max;
  wa.max_concentration = o->max_concentration;
  wa.out_mode = static_cast<double*>(out_mode);
  wa.out_conc = out_concentration;
  TimedRegion tr(h, as_stream(stream));
  return pbbss::cw_launch(D, K, o->y_is_c128, wa, h->cfg, as_stream(stream));
}

PBBSS_API int pbbss_wmwf(pbbss_handle_t h, const void* target, const void* noise, int64_t N, int D,
                         double distortion_weight, int frequency_dependent, void* out_mat,
                         void* out_snr_num, void* out_snr_den, int32_t* out_status,
                         void* stream) {
  DeviceGuard device_guard(h);
  if (!h || !target || !noise || !out_mat || N <= 0) return PBBSS_ERR_INVALID_ARG;
  if (D > 8)
    return pbbss::launch_gen_souden(static_cast<const double*>(target),
                                    static_cast<const double*>(noise), N, D, distortion_weight,
                                    frequency_dependent ? 2 : 1, static_cast<double*>(out_mat),
                                    static_cast<double*>(out_snr_num),
                                    static_cast<double*>(out_snr_den), out_status,
                                    h->cfg.lds_limit, as_stream(stream));
  return pbbss::launch_mvdr_souden(static_cast<const double*>(target),
                                   static_cast<const double*>(noise), N, D, distortion_weight,
                                   frequency_dependent ? 2 : 1, static_cast<double*>(out_mat),
                                   static_cast<double*>(out_snr_num),
                                   static_cast<double*>(out_snr_den), out_status,
                                   as_stream(stream));
}

// ---------------------------------------------------------------------------
// N2/N3: real-embedding mixtures and the joint spatial+spectral models.  These
// are multi-kernel loops enqueued asynchronously on `stream` (no host sync).
// ---------------------------------------------------------------------------
namespace {
inline bool embed_shape_ok(int64_t B, int64_t N, int E, int K) {
  return B >= 1 && B <= 65535 && N >= 1 && E >= 1 && E <= pbbss::kEmbedMaxE && K >= 1 &&
         K <= pbbss::kEmbedMaxK;
}
}  // namespace

PBBSS_API int pbbss_embed_log_pdf(pbbss_handle_t h, const void* y, int y_is_f64, int64_t B,
                                  int64_t N, int E, int K, int kind, const double* mean,
                                  const double* scale, double* out_log_pdf, void* stream) {
  DeviceGuard device_guard(h);
  if (!h || !y || !mean || !scale || !out_log_pdf) return PBBSS_ERR_INVALID_ARG;
  if (!embed_shape_ok(B, N, E, K)) return PBBSS_ERR_UNSUPPORTED;
  hipStream_t s = as_stream(stream);
  const size_t esz = y_is_f64 ? 8 : 4;
  const size_t need = WorkCarver::pad((size_t)B * E * N * esz) + 2 * WorkCarver::pad((size_t)B * K * 8);
  void* w = handle_work(h, need);
  if (!w) return PBBSS_ERR_HIP;
  WorkCarver wc(w);
  char* yd = wc.take<char>((size_t)B * E * N * esz);
  double* offset = wc.take<double>((size_t)B * K);
  double* prec = wc.take<double>((size_t)B * K);
  int rc = pbbss::launch_embed_prepare(y, y_is_f64, B, N, E, 0, yd, nullptr, s);
  if (rc != PBBSS_OK) return rc;
  if (kind == PBBSS_EMBED_GAUSS_DIAG) {
    if (B != 1) return PBBSS_ERR_UNSUPPORTED;  // the reference's DiagonalGaussian has no batch axis
    void* cw = handle_scratch(h, pbbss::diag_consts_doubles(K, E) * 8);
    if (!cw) return PBBSS_ERR_HIP;
    return pbbss::launch_diag_estep(yd, y_is_f64, N, E, K, mean, scale, 1.0, N,
                                    static_cast<double*>(cw), out_log_pdf, s);
  }
  rc = pbbss::launch_embed_offsets(kind, B * K, E, scale, offset, prec, s);
  if (rc != PBBSS_OK) return rc;
  return pbbss::launch_embed_estep(kind, yd, y_is_f64, B, N, E, K, mean, prec, offset, nullptr,
                                   1.0, N, out_log_pdf, nullptr, s);
}

PBBSS_API int pbbss_estimate_mixture_weight(pbbss_handle_t h, const double* affiliation,
                                            const double* saliency, int64_t Bo, int64_t Bi, int K,
                                            int64_t N, int reduce_inner, int reduce_n,
                                            double* out_weight, void* stream) {
  DeviceGuard device_guard(h);
  if (!h || !affiliation || !out_weight || Bo <= 0 || Bi <= 0 || N <= 0) return PBBSS_ERR_INVALID_ARG;
  if (K < 1 || K > 64 || (saliency && K > 16)) return PBBSS_ERR_UNSUPPORTED;
  void* w = handle_work(h, WorkCarver::pad(pbbss::mixture_weight_tmp_doubles(Bo, Bi, K, N, reduce_n) * 8));
  if (!w) return PBBSS_ERR_HIP;
  return pbbss::launch_mixture_weight(affiliation, saliency, Bo, Bi, K, N, reduce_inner ? 1 : 0,
                                      reduce_n ? 1 : 0, static_cast<double*>(w), out_weight,
                                      as_stream(stream));
}

PBBSS_API int pbbss_log_pdf_to_affiliation(pbbss_handle_t h, const double* log_pdf, int64_t B, int K,
                                           int64_t N, const double* weight, int64_t wb, int64_t wk,
                                           int64_t wn, const uint8_t* activity,
                                           double affiliation_eps, double* out_affiliation,
                                           void* stream) {
  DeviceGuard device_guard(h);
  if (!h || !log_pdf || !weight || !out_affiliation || B <= 0 || N <= 0 || K < 1)
    return PBBSS_ERR_INVALID_ARG;
  if (wb < 0 || wk < 0 || wn < 0) return PBBSS_ERR_INVALID_ARG;
  return pbbss::launch_log_pdf_to_affiliation(log_pdf, B, K, N, weight, wb, wk, wn, activity,
                                              affiliation_eps, out_affiliation, as_stream(stream));
}

PBBSS_API int pbbss_log_pdf_to_affiliation_inline_pa(
    pbbss_handle_t h, const double* spatial_log_pdf, const double* spectral_log_pdf, int64_t F, int K,
    int64_t T, const double* weight, int64_t wb, int64_t wk, int64_t wn, const uint8_t* activity,
    double affiliation_eps, double* out_affiliation, int32_t* out_permutation, void* stream) {
  DeviceGuard device_guard(h);
  if (!h || !spatial_log_pdf || !spectral_log_pdf || !weight || !out_affiliation || F <= 0 ||
      T <= 0 || K < 1)
    return PBBSS_ERR_INVALID_ARG;
  if (wb < 0 || wk < 0 || wn < 0) return PBBSS_ERR_INVALID_ARG;
  if (T > 2147483647LL) return PBBSS_ERR_UNSUPPORTED;
  // the permutation search of the generic-size joint E-step without its M-step half: no
  // observation, no quadratic forms (K > 6: 5 040+ permutations per bin -- refused)
  return pbbss::launch_gen_joint_pa(nullptr, 0, F, (int)T, 0, K, spatial_log_pdf, nullptr,
                                    spectral_log_pdf, 1.0, weight, wb, wk, wn, nullptr,
                                    affiliation_eps, out_affiliation, nullptr, nullptr,
                                    as_stream(stream), activity, out_permutation);
}

PBBSS_API int pbbss_embed_fit(pbbss_handle_t h, const void* y, int y_is_f64, int64_t B, int64_t N,
                              int E, int K, int kind, int normalize, const double* weights,
                              double min_concentration, double max_concentration,
                              double* out_mean, double* out_scale, void* stream) {
  DeviceGuard device_guard(h);
  if (!h || !y || !weights || !out_mean || !out_scale) return PBBSS_ERR_INVALID_ARG;
  if (!embed_shape_ok(B, N, E, K)) return PBBSS_ERR_UNSUPPORTED;
  hipStream_t s = as_stream(stream);
  const size_t np = pbbss::embed_partial_doubles(B, N, E, K, nullptr);
  size_t need = WorkCarver::pad(np * 8);
  if (normalize) need += 2 * WorkCarver::pad((size_t)B * N * E * 8);
  void* w = handle_work(h, need);
  if (!w) return PBBSS_ERR_HIP;
  WorkCarver wc(w);
  double* part = wc.take<double>(np);
  const void* yr = y;
  int yr_f64 = y_is_f64;
  if (normalize) {
    double* yd = wc.take<double>((size_t)B * N * E);
    double* yn = wc.take<double>((size_t)B * N * E);
    int rc = pbbss::launch_embed_prepare(y, y_is_f64, B, N, E, 1, yd, yn, s);
    if (rc != PBBSS_OK) return rc;
    yr = yn;
    yr_f64 = 1;
  }
  return pbbss::launch_embed_fit(kind, yr, yr_f64, B, N, E, K, weights, N, nullptr,
                                 min_concentration, max_concentration, -1, part, out_mean,
                                 out_scale, nullptr, nullptr, nullptr, 0, s);
}

PBBSS_API int pbbss_gauss_full_fit(pbbss_handle_t h, const void* y, int y_is_f64, int64_t B,
                                   int64_t N, int E, int K, const double* weights,
                                   double* out_mean, double* out_covariance, void* stream) {
  DeviceGuard device_guard(h);
  if (!h || !y || !weights || !out_mean || !out_covariance || B <= 0 || N <= 0)
    return PBBSS_ERR_INVALID_ARG;
  if (E < 1 || E > pbbss::kGaussFullMaxE || K < 1) return PBBSS_ERR_UNSUPPORTED;
  const size_t np = pbbss::gauss_full_partial_doubles(B, N, E, K);
  void* w = handle_work(h, WorkCarver::pad(np * 8));
  if (!w) return PBBSS_ERR_HIP;
  TimedRegion tr(h, as_stream(stream));
  return pbbss::launch_gauss_full_fit(y, y_is_f64, B, N, E, K, weights, nullptr,
                                      static_cast<double*>(w), out_mean, out_covariance, nullptr,
                                      nullptr, nullptr, nullptr, as_stream(stream));
}

PBBSS_API int pbbss_gauss_full_log_pdf(pbbss_handle_t h, const void* y, int y_is_f64, int64_t B,
                                       int64_t N, int E, int K, const double* mean,
                                       const double* covariance, double* out_log_pdf,
                                       int32_t* out_status, void* stream) {
  DeviceGuard device_guard(h);
  if (!h || !y || !mean || !covariance || !out_log_pdf || !out_status || B <= 0 || N <= 0)
    return PBBSS_ERR_INVALID_ARG;
  if (E < 1 || E > pbbss::kGaussFullMaxE || K < 1 || K > 64) return PBBSS_ERR_UNSUPPORTED;
  const size_t nm = (size_t)B * K * E * E;
  void* w = handle_work(h, WorkCarver::pad(nm * 8) + WorkCarver::pad((size_t)B * K * 8));
  if (!w) return PBBSS_ERR_HIP;
  WorkCarver wc(w);
  double* mq = wc.take<double>(nm);
  double* off = wc.take<double>((size_t)B * K);
  hipStream_t s = as_stream(stream);
  TimedRegion tr(h, s);
  int rc = pbbss::launch_gauss_full_factor(covariance, B * K, E, mq, off, out_status, s);
  if (rc != PBBSS_OK) return rc;
  return pbbss::launch_gauss_full_logpdf(y, y_is_f64, B, N, E, K, mean, mq, off, nullptr,
                                         out_log_pdf, nullptr, s);
}

PBBSS_API int pbbss_gmm_full_fit(pbbss_handle_t h, const void* y, int64_t B, int64_t N, int E,
                                 int K, const double* gamma0, const double* in_mean,
                                 const double* in_covariance, const double* in_weight,
                                 const double* saliency, const double* fixed_covariance,
                                 const pbbss_mix_opts* o, double* out_mean,
                                 double* out_covariance, double* out_weight,
                                 double* out_affiliation, double* out_log_pdf,
                                 int32_t* out_status, void* stream) {
  DeviceGuard device_guard(h);
  if (!h || !y || !o || !out_status || B <= 0 || N <= 0) return PBBSS_ERR_INVALID_ARG;
  if (E < 1 || E > pbbss::kGaussFullMaxE || K < 1 || K > 64) return PBBSS_ERR_UNSUPPORTED;
  if (o->iterations < 0 || o->weight_mode < 0 || o->weight_mode > 1) return PBBSS_ERR_INVALID_ARG;
  const bool has_gamma = gamma0 != nullptr;
  const bool has_model = in_mean && in_covariance && in_weight;
  if (has_gamma == has_model) return PBBSS_ERR_INVALID_ARG;
  if ((o->iterations == 0) != has_model) return PBBSS_ERR_INVALID_ARG;
  if (!out_mean || !out_covariance || !out_weight) return PBBSS_ERR_INVALID_ARG;
  hipStream_t s = as_stream(stream);
  const size_t np = pbbss::gauss_full_partial_doubles(B, N, E, K);
  const size_t nm = (size_t)B * K * E * E;
  void* w = handle_work(h, WorkCarver::pad(np * 8) + WorkCarver::pad(nm * 8) +
                               WorkCarver::pad((size_t)B * K * N * 8) +
                               2 * WorkCarver::pad((size_t)B * K * 8));
  if (!w) return PBBSS_ERR_HIP;
  WorkCarver wc(w);
  double* part = wc.take<double>(np);
  double* mq = wc.take<double>(nm);
  double* aff = wc.take<double>((size_t)B * K * N);
  double* off = wc.take<double>((size_t)B * K);
  double* s0 = wc.take<double>((size_t)B * K);
  const int f64 = o->embedding_is_f64;
  TimedRegion tr(h, s);
  int rc;
  if (has_model) {
    if ((rc = copy_d2d(out_mean, in_mean, (size_t)B * K * E * 8, s)) != PBBSS_OK) return rc;
    if ((rc = copy_d2d(out_covariance, in_covariance, nm * 8, s)) != PBBSS_OK) return rc;
    if ((rc = copy_d2d(out_weight, in_weight, (size_t)B * K * 8, s)) != PBBSS_OK) return rc;
    rc = pbbss::launch_gauss_full_factor(out_covariance, B * K, E, mq, off, out_status, s);
    if (rc != PBBSS_OK) return rc;
  }
  for (int it = 0; it < o->iterations; ++it) {
    const double* src = gamma0;
    if (it > 0) {  // gmm.py:129-130
      rc = pbbss::launch_gauss_full_logpdf(y, f64, B, N, E, K, out_mean, mq, off, out_weight,
                                           nullptr, aff, s);
      if (rc != PBBSS_OK) return rc;
      src = aff;
    }
    rc = pbbss::launch_gauss_full_fit(y, f64, B, N, E, K, src, saliency, part, out_mean,
                                      out_covariance, fixed_covariance ? nullptr : mq,
                                      fixed_covariance ? nullptr : off, s0, out_status, s);
    if (rc != PBBSS_OK) return rc;
    rc = pbbss::launch_gauss_full_weights(s0, B, K, o->weight_mode, out_weight, s);
    if (rc != PBBSS_OK) return rc;
    if (fixed_covariance) {  // gmm.py:160-167
      if ((rc = copy_d2d(out_covariance, fixed_covariance, nm * 8, s)) != PBBSS_OK) return rc;
      rc = pbbss::launch_gauss_full_factor(out_covariance, B * K, E, mq, off, out_status, s);
      if (rc != PBBSS_OK) return rc;
    }
  }
  if (o->final_predict && (out_affiliation || out_log_pdf)) {
    rc = pbbss::launch_gauss_full_logpdf(y, f64, B, N, E, K, out_mean, mq, off, out_weight,
                                         out_log_pdf, out_affiliation, s);
    if (rc != PBBSS_OK) return rc;
  }
  return PBBSS_OK;
}

// EM loop shared by the two real-embedding mixtures: kind = PBBSS_EMBED_VMF (rows unit-normalised
// first, vmfmm.py:76-78) or PBBSS_EMBED_GAUSS_SPHERICAL (gmm.py:126-171, covariance_type
// 'spherical'; fixed_scale = fixed_covariance (B,K) or null).
static int embed_mixture_fit(pbbss_handle_t h, int kind, const void* y, int64_t B, int64_t N, int E,
                             int K, const double* gamma0, const double* in_mean,
                             const double* in_scale, const double* in_weight,
                             const double* saliency, const double* fixed_scale,
                             const pbbss_mix_opts* o, double* out_mean, double* out_scale,
                             double* out_weight, double* out_affiliation, double* out_log_pdf,
                             void* stream) {
  if (!h || !y || !o) return PBBSS_ERR_INVALID_ARG;
  if (!embed_shape_ok(B, N, E, K)) return PBBSS_ERR_UNSUPPORTED;
  if (o->iterations < 0 || o->weight_mode < 0 || o->weight_mode > 1) return PBBSS_ERR_INVALID_ARG;
  const bool has_gamma = gamma0 != nullptr;
  const bool has_model = in_mean && in_scale && in_weight;
  if (has_gamma == has_model) return PBBSS_ERR_INVALID_ARG;
  if ((o->iterations == 0) != has_model) return PBBSS_ERR_INVALID_ARG;  // vmfmm.py:70-74
  if (!out_mean || !out_scale || !out_weight) return PBBSS_ERR_INVALID_ARG;
  const bool vmf = (kind == PBBSS_EMBED_VMF);
  hipStream_t s = as_stream(stream);
  // many small vMF mixtures (e.g. one per frequency bin): the persistent one-workgroup-per-mixture
  // kernel runs the whole loop in ONE launch (embed.hip: vmf_bin_em_kernel); a big mixture is
  // better off spread over the chip by the sweep + finalize pair below
  if (vmf && !out_log_pdf && B >= 16) {
    const size_t lds = pbbss::vmf_bin_lds_bytes(N, E, K, o->embedding_is_f64);
    if (lds > 0 && lds <= h->cfg.lds_limit) {
      if (has_model) {
        int rc0;
        if ((rc0 = copy_d2d(out_mean, in_mean, (size_t)B * K * E * 8, s)) != PBBSS_OK) return rc0;
        if ((rc0 = copy_d2d(out_scale, in_scale, (size_t)B * K * 8, s)) != PBBSS_OK) return rc0;
        if ((rc0 = copy_d2d(out_weight, in_weight, (size_t)B * K * 8, s)) != PBBSS_OK) return rc0;
      }
      TimedRegion tr(h, s);
      const int rc2 = pbbss::launch_vmf_bin_em2(
          y, o->embedding_is_f64, B, N, E, K, o->iterations, gamma0, saliency,
          has_model ? out_mean : nullptr, has_model ? out_scale : nullptr,
          has_model ? out_weight : nullptr, o->min_concentration, o->max_concentration,
          o->weight_mode, out_mean, out_scale, out_weight,
          (o->final_predict && out_affiliation) ? out_affiliation : nullptr, h->cfg.lds_limit,
          h->cfg.num_cu > 0 ? h->cfg.num_cu : 256, s);
      if (rc2 != PBBSS_ERR_UNSUPPORTED) return rc2;
      return pbbss::launch_vmf_bin_em(
          y, o->embedding_is_f64, B, N, E, K, o->iterations, gamma0, saliency,
          has_model ? out_mean : nullptr, has_model ? out_scale : nullptr,
          has_model ? out_weight : nullptr, o->min_concentration, o->max_concentration,
          o->weight_mode, out_mean, out_scale, out_weight,
          (o->final_predict && out_affiliation) ? out_affiliation : nullptr, h->cfg.lds_limit, s);
    }
  }
  const size_t nyz = (size_t)B * N * E;
  const size_t np = pbbss::embed_partial_doubles(B, N, E, K, nullptr);
  // vMF mixture: ONE pass over the embedding per iteration (vmf_em_kernel, embed.hip) where the
  // LDS tile fits; the two-kernel path (transposed copy for the E-step, row-major sweep for the
  // M-step) otherwise, for the Gaussian mixture and for the log-pdf output
  const size_t npf = vmf ? pbbss::vmf_fused_partial_doubles(B, N, E, K, o->embedding_is_f64) : 0;
  const bool fused = npf > 0;
  const bool need_copy = !fused || (o->final_predict && out_log_pdf);
  const size_t need = (need_copy ? WorkCarver::pad(nyz * 8) + WorkCarver::pad((size_t)B * N * 8) : 0) +
                      (fused ? 0 : WorkCarver::pad((size_t)B * K * N * 8)) +
                      WorkCarver::pad((np > npf ? np : npf) * 8) +
                      2 * WorkCarver::pad((size_t)B * K * 8);
  void* w = handle_work(h, need);
  if (!w) return PBBSS_ERR_HIP;
  WorkCarver wc(w);
  double* yd = need_copy ? wc.take<double>(nyz) : nullptr;  // (B,E,N) transposed copy, INPUT type
  double* rowscale = need_copy ? wc.take<double>((size_t)B * N) : nullptr;  // vMF: 1 / |y_n|
  double* aff = fused ? nullptr : wc.take<double>((size_t)B * K * N);
  double* part = wc.take<double>(np > npf ? np : npf);
  double* offset = wc.take<double>((size_t)B * K);
  double* prec = wc.take<double>((size_t)B * K);
  // Two-kernel path: the E-step reads the transposed copy, the M-step the caller's row-major
  // array, both in the caller's type.  The vMF mixture works on unit rows: the E-step normalises
  // its dot products itself, the M-step takes 1 / |y_n| from `rowscale`.
  const int e_f64 = o->embedding_is_f64;
  const void* fit_y = y;
  TimedRegion tr(h, s);
  int rc = PBBSS_OK;
  if (need_copy) {
    rc = pbbss::launch_embed_prepare(y, o->embedding_is_f64, B, N, E, 0, yd, nullptr, s,
                                     vmf ? rowscale : nullptr);
    if (rc != PBBSS_OK) return rc;
  }
  if (has_model) {
    if ((rc = copy_d2d(out_mean, in_mean, (size_t)B * K * E * 8, s)) != PBBSS_OK) return rc;
    if ((rc = copy_d2d(out_scale, in_scale, (size_t)B * K * 8, s)) != PBBSS_OK) return rc;
    if ((rc = copy_d2d(out_weight, in_weight, (size_t)B * K * 8, s)) != PBBSS_OK) return rc;
  }
  for (int it = 0; it < o->iterations; ++it) {
    if (fused) {  // vmfmm.py:131-172: E-step with the previous model (or the initialisation), M-step
      rc = pbbss::launch_vmf_em(y, e_f64, B, N, E, K, it == 0 ? gamma0 : nullptr, saliency,
                                o->min_concentration, o->max_concentration, o->weight_mode, part,
                                out_mean, out_scale, out_weight, offset, prec, nullptr, 1, s);
      if (rc != PBBSS_OK) return rc;
      continue;
    }
    const double* src = gamma0;
    if (it > 0) {  // vmfmm.py:137-138 / gmm.py:129-130 (offsets: previous M-step's finalize)
      rc = pbbss::launch_embed_estep(kind, yd, e_f64, B, N, E, K, out_mean, prec, offset,
                                     out_weight, 1.0, N, nullptr, aff, s);
      if (rc != PBBSS_OK) return rc;
      src = aff;
    }
    rc = pbbss::launch_embed_fit(kind, fit_y, e_f64, B, N, E, K, src, N, saliency,
                                 o->min_concentration, o->max_concentration, o->weight_mode, part,
                                 out_mean, out_scale, out_weight, offset, prec, 0, s,
                                 vmf ? rowscale : nullptr);
    if (rc != PBBSS_OK) return rc;
    if (fixed_scale) {  // gmm.py:160-167
      if ((rc = copy_d2d(out_scale, fixed_scale, (size_t)B * K * 8, s)) != PBBSS_OK) return rc;
      rc = pbbss::launch_embed_offsets(kind, B * K, E, out_scale, offset, prec, s);
      if (rc != PBBSS_OK) return rc;
    }
  }
  if (o->final_predict && (out_affiliation || out_log_pdf)) {
    if (o->iterations == 0) {
      rc = pbbss::launch_embed_offsets(kind, B * K, E, out_scale, offset, prec, s);
      if (rc != PBBSS_OK) return rc;
    }
    if (fused && !out_log_pdf) {
      return pbbss::launch_vmf_em(y, e_f64, B, N, E, K, nullptr, nullptr, o->min_concentration,
                                  o->max_concentration, o->weight_mode, part, out_mean, out_scale,
                                  out_weight, offset, prec, out_affiliation, 0, s);
    }
    rc = pbbss::launch_embed_estep(kind, yd, e_f64, B, N, E, K, out_mean, prec, offset,
                                   out_weight, 1.0, N, out_log_pdf, out_affiliation, s);
    if (rc != PBBSS_OK) return rc;
  }
  return PBBSS_OK;
}

PBBSS_API int pbbss_vmfmm_fit(pbbss_handle_t h, const void* y, int64_t B, int64_t N, int E, int K,
                              const double* gamma0, const double* in_mean,
                              const double* in_concentration, const double* in_weight,
                              const double* saliency, const pbbss_mix_opts* o, double* out_mean,
                              double* out_concentration, double* out_weight,
                              double* out_affiliation, double* out_log_pdf, void* stream) {
  DeviceGuard device_guard(h);
  return embed_mixture_fit(h, PBBSS_EMBED_VMF, y, B, N, E, K, gamma0, in_mean, in_concentration,
                           in_weight, saliency, nullptr, o, out_mean, out_concentration,
                           out_weight, out_affiliation, out_log_pdf, stream);
}

PBBSS_API int pbbss_gmm_fit(pbbss_handle_t h, const void* y, int64_t B, int64_t N, int E, int K,
                            const double* gamma0, const double* in_mean,
                            const double* in_covariance, const double* in_weight,
                            const double* saliency, const double* fixed_covariance,
                            const pbbss_mix_opts* o, double* out_mean, double* out_covariance,
                            double* out_weight, double* out_affiliation, double* out_log_pdf,
                            void* stream) {
  DeviceGuard device_guard(h);
  return embed_mixture_fit(h, PBBSS_EMBED_GAUSS_SPHERICAL, y, B, N, E, K, gamma0, in_mean,
                           in_covariance, in_weight, saliency, fixed_covariance, o, out_mean,
                           out_covariance, out_weight, out_affiliation, out_log_pdf, stream);
}

// dynamic LDS of the joint kernels for (D, K, T): EmKernel<D,K,YS,false>::lds_bytes(T) + the
// 64 bytes of the inline aligner's class permutation, written out for a run-time D
static size_t joint_kernel_lds_bytes(int D, int K, int T, int c128) {
  // frame arrays in chunks of 64 frames (cacgmm_em.hpp: EmKernel::padded_frames)
  const size_t DP = (size_t)(D + 1) / 2, Tp = (size_t)((T + 63) & ~63), NA = (size_t)D * D;
  const size_t frames = DP * Tp * 4 * (c128 ? 8 : 4) + Tp * 8 + (size_t)K * Tp * 8;
  // (+ the write-back table of the M phase and its sink: EmKernel::wbtab_bytes() + 8)
  const size_t noff = (size_t)D * (D - 1) / 2, nslot = (D + 3) / 4 + 2 * ((noff + 3) / 4);
  const size_t wbtab = (4 * 16 * ((K * nslot + 15) / 16) * 2 + 7) & ~(size_t)7;
  const size_t small = 2 * (size_t)K * NA * 8 + (size_t)K * 8 * 4 + 4 * (size_t)K * 8 +
                       (size_t)K * 4 * 2 + 16 + wbtab + 8;
  return ((frames + small + 15) & ~(size_t)15) + 64;
}

// helper blocks of the in-launch spectral finalize of the rotated joint loop (embed_dev.hpp)
static constexpr int kJointFinHelpers = 32;
static int joint_fin_helpers() {  // development knob: helper blocks actually launched (<= 32)
  static const int n = [] {
    const char* v = getenv("PBBSS_JOINT_FIN_HELPERS");
    const int x = v ? atoi(v) : kJointFinHelpers;
    return x < 1 ? 1 : (x > kJointFinHelpers ? kJointFinHelpers : x);
  }();
  return n;
}

PBBSS_API int pbbss_joint_fit(pbbss_handle_t h, const void* observation, const void* embedding,
                              int64_t F, int T, int D, int E, int K, const double* gamma0,
                              const void* in_eigvec, const double* in_eigval,
                              const double* in_weight, const double* in_mean,
                              const double* in_scale, const double* saliency,
                              const pbbss_mix_opts* o, void* out_eigvec, double* out_eigval,
                              double* out_weight, double* out_mean, double* out_scale,
                              int32_t* out_status, double* out_affiliation, void* stream) {
  DeviceGuard device_guard(h);
  // (no residency gate: the member workgroups of the joint kernels never wait for each other --
  // the last arriver finishes the problem, run_joint_member in cacgmm_em.hpp)
  if (!h || !observation || !embedding || !o || F <= 0 || T <= 0) return PBBSS_ERR_INVALID_ARG;
  // 9 <= D <= 32 or 7..8 classes: the spatial half runs on the generic-size kernels
  // (generic.hip), one E-step and one M-step launch group per iteration around the same spectral
  // kernels
  // ... and so does an utterance too long for the LDS-resident joint kernels (no HBM-scratch
  // variant of those): the generic kernels stream the frames (round 4; the reference has no
  // length limit)
  const size_t joint_lds =
      D <= 8 ? joint_kernel_lds_bytes(D, K, T, o->obs_is_c128) : (size_t)0;
  const bool gen = D > 8 || K > 6 || joint_lds > h->cfg.lds_limit;
  if (D < 2 || K < 1 || K > pbbss::kEmbedMaxK || (gen && !pbbss::gen_supported(D, K)))
    return PBBSS_ERR_UNSUPPORTED;
  if (gen && (F > 65535 || (o->inline_pa && K > 6))) return PBBSS_ERR_UNSUPPORTED;
  const int64_t N = F * (int64_t)T;
  if (!embed_shape_ok(1, N, E, K)) return PBBSS_ERR_UNSUPPORTED;
  if (o->iterations < 0 || o->weight_mode < 0 || o->weight_mode > 4) return PBBSS_ERR_INVALID_ARG;
  if (o->kind < PBBSS_EMBED_VMF || o->kind > PBBSS_EMBED_GAUSS_DIAG) return PBBSS_ERR_UNSUPPORTED;
  const bool g_full = o->kind == PBBSS_EMBED_GAUSS_FULL, g_diag = o->kind == PBBSS_EMBED_GAUSS_DIAG;
  if (g_full && E > pbbss::kGaussFullMaxE) return PBBSS_ERR_UNSUPPORTED;
  // opts->sharded: this call holds ONE RANK'S BLOCK of the frequency bins; the spectral M-step
  // sums and the bin-constant class weights are summed over the communicator of the handle, in
  // stream order (no host round trip inside the loop).  The full-covariance scatter centres its
  // augmented vectors on a shift that must be THE SAME on every rank for the Gram tiles to add up:
  // rank 0's first embedding row, broadcast once per fit by an all-reduce (the others contribute
  // zeros); the reduced tiles are all-reduced between the reduction and the finalize kernel.
  const bool sharded = o->sharded != 0 && o->iterations > 0;
  if (sharded && !h->comm) return PBBSS_ERR_INVALID_ARG;  // pbbss_comm_create first
  pbbss::PartialReduce all_ranks{
      [](void* ctx, double* buf, size_t count, hipStream_t st) -> int {
        return pbbss::comm_all_reduce_f64(static_cast<pbbss_handle_t>(ctx)->comm, buf, count, st);
      },
      h};
  const pbbss::PartialReduce* reduce = sharded ? &all_ranks : nullptr;
  // scalars per class of the spectral model's second parameter: concentration / variance (1),
  // per-dimension variances (E), covariance matrix (E * E)
  const size_t nscale = g_full ? (size_t)K * E * E : (g_diag ? (size_t)K * E : (size_t)K);
  if (o->covariance_norm < 0 || o->covariance_norm > 2) return PBBSS_ERR_INVALID_ARG;
  const bool has_gamma = gamma0 != nullptr;
  const bool has_model = in_eigvec && in_eigval && in_weight && in_mean && in_scale;
  if (has_gamma == has_model) return PBBSS_ERR_INVALID_ARG;
  if ((o->iterations == 0) != has_model) return PBBSS_ERR_INVALID_ARG;
  if (!out_eigvec || !out_eigval || !out_weight || !out_mean || !out_scale || !out_status)
    return PBBSS_ERR_INVALID_ARG;
  hipStream_t s = as_stream(stream);
  int64_t wb = 0, wk = 0, wt = 0;
  size_t wcount = 1;
  switch (o->weight_mode) {
    case PBBSS_JOINT_WEIGHT_FK: wb = K; wk = 1; wcount = (size_t)F * K; break;
    case PBBSS_JOINT_WEIGHT_K: wk = 1; wcount = K; break;
    case PBBSS_JOINT_WEIGHT_KT: wk = T; wt = 1; wcount = (size_t)K * T; break;
    default: break;
  }
  const size_t esz = o->embedding_is_f64 ? 8 : 4;
  // Rotated loop (round 4): ONE pass over the embedding per EM iteration -- the sweep kernel forms
  // the posteriors from the spatial quadratic forms Q and the spectral log-pdf of the SAME tile of
  // embedding rows it then accumulates the spectral M-step sums from, the spatial kernel runs
  // M-step, factorisation and the NEXT model's quadratic forms (embed.hip: joint_sweep_kernel,
  // cacgmm_em.hpp: run_joint_ms).  Served: vMF / spherical Gaussian, D <= 8, K <= 6, no inline
  // aligner, no fixed covariance; everything else keeps the three-kernel path below.
  // PBBSS_JOINT_ROTATED=0 switches it off (A/B runs, tests of the other path).
  static const bool rot_allowed = [] {
    const char* v = getenv("PBBSS_JOINT_ROTATED");
    return !(v && v[0] == '0');
  }();
  const bool rot = rot_allowed && !gen && !o->inline_pa && !(in_scale && gamma0) &&
                   o->iterations >= 2 &&
                   pbbss::joint_sweep_supported(o->kind, N, E, K, o->embedding_is_f64);
  const size_t np0 = pbbss::embed_partial_doubles(1, N, E, K, nullptr);
  const size_t npj = rot ? pbbss::joint_sweep_partial_doubles(o->kind, N, E, K, o->embedding_is_f64) : 0;
  const size_t np = np0 > npj ? np0 : npj;
  const size_t nfkt = (size_t)F * K * T;
  const size_t nstate = (size_t)F * K * (D * D + 2);
  const size_t ngp = g_full ? pbbss::gauss_full_partial_doubles(1, N, E, K) : 0;
  const size_t nconst = g_diag ? pbbss::diag_consts_doubles(K, E) : 0;
  const size_t nmat = (size_t)F * K;
  const size_t ntmp = pbbss::joint_weight_tmp_doubles(o->weight_mode, F, K, T);
  const size_t ninv = gen ? pbbss::gen_state_doubles((int64_t)nmat, D) : 0;
  const size_t nyt = gen ? (size_t)F * T * D * (o->obs_is_c128 ? 16 : 8) : 0;
  const size_t need_gen =
      gen ? (o->inline_pa ? 3 : 1) * WorkCarver::pad(nfkt * 8) + WorkCarver::pad(nmat * D * D * 16) +
                WorkCarver::pad(ninv * 8) +
                2 * WorkCarver::pad(nmat * 8) + WorkCarver::pad((size_t)F * 4) + WorkCarver::pad(nyt)
          : 0;
  const size_t need = need_gen + WorkCarver::pad((size_t)E * N * esz) + 2 * WorkCarver::pad(nfkt * 8) +
                      WorkCarver::pad(np * 8) + 2 * WorkCarver::pad((size_t)K * 8) +
                      WorkCarver::pad(ntmp * 8) + WorkCarver::pad(nstate * 8) +
                      (g_full ? 2 * WorkCarver::pad(nfkt * 8) + WorkCarver::pad(ngp * 8) +
                                    WorkCarver::pad((size_t)K * E * E * 8)
                              : 0) +
                      WorkCarver::pad(nconst * 8) + WorkCarver::pad(64) +
                      WorkCarver::pad((size_t)F * K * 8) +
                      WorkCarver::pad((size_t)kJointFinHelpers * 2 * K * (E + 1) * 8) +
                      WorkCarver::pad((size_t)E * 8);
  void* w = handle_work(h, need);
  if (!w) return PBBSS_ERR_HIP;
  WorkCarver wc(w, need);
  char* yd = wc.take<char>((size_t)E * N * esz);
  double* aff = wc.take<double>(nfkt);
  double* slp = wc.take<double>(nfkt);
  double* part = wc.take<double>(np);
  double* offset = wc.take<double>(K);
  double* prec = wc.take<double>(K);
  double* tmp = wc.take<double>(ntmp);
  double* jstate = wc.take<double>(nstate);
  double* wkn = g_full ? wc.take<double>(nfkt) : nullptr;    // (K, F*T) class weights
  double* lpkn = g_full ? wc.take<double>(nfkt) : nullptr;   // (K, F*T) log-pdf
  double* gpart = g_full ? wc.take<double>(ngp) : nullptr;
  double* mq = g_full ? wc.take<double>((size_t)K * E * E) : nullptr;
  double* dconst = g_diag ? wc.take<double>(nconst) : nullptr;
  int32_t* gst = reinterpret_cast<int32_t*>(wc.take<char>(64));  // status of the spectral half
  double* lndet = wc.take<double>((size_t)F * K);  // rotated loop: ln det B_fk of the current model
  double* fin_tmp = wc.take<double>((size_t)kJointFinHelpers * 2 * K * (E + 1));
  double* gshift = wc.take<double>((size_t)E);  // sharded full covariance: the common shift
  // generic-size spatial half: M-step weights, covariances, inverse state, class sums, zero-frame
  // flags, frame-contiguous copy of the observation
  double* g_mw = gen ? wc.take<double>(nfkt) : nullptr;
  double* g_cov = gen ? wc.take<double>(nmat * D * D * 2) : nullptr;
  double* g_inv = gen ? wc.take<double>(ninv) : nullptr;
  double* g_logdet = gen ? wc.take<double>(nmat) : nullptr;
  double* g_csum = gen ? wc.take<double>(nmat) : nullptr;
  int32_t* g_zero = gen ? wc.take<int32_t>((size_t)F) : nullptr;
  char* g_yt = gen ? wc.take<char>(nyt) : nullptr;
  double* g_lp = (gen && o->inline_pa) ? wc.take<double>(nfkt) : nullptr;  // spatial log-pdf
  double* g_q = (gen && o->inline_pa) ? wc.take<double>(nfkt) : nullptr;   // quadratic forms
  if (!wc.fits()) return PBBSS_ERR_INTERNAL;
  if (hipMemsetAsync(gst, 0, 64, as_stream(stream)) != hipSuccess) return PBBSS_ERR_HIP;
  TimedRegion tr(h, s);
  int rc = pbbss::launch_embed_prepare(embedding, o->embedding_is_f64, 1, N, E, 0, yd, nullptr, s);
  if (rc != PBBSS_OK) return rc;
  if (sharded && g_full) {
    // rank 0's first embedding row (converted to float64) on every rank
    if (h->comm_rank == 0) {
      rc = pbbss::launch_first_row_f64(embedding, o->embedding_is_f64, E, gshift, s);
      if (rc != PBBSS_OK) return rc;
    } else if (hipMemsetAsync(gshift, 0, (size_t)E * 8, s) != hipSuccess) {
      return PBBSS_ERR_HIP;
    }
    if ((rc = all_ranks.fn(all_ranks.ctx, gshift, (size_t)E, s)) != PBBSS_OK) return rc;
  }
  if (has_model) {
    if ((rc = copy_d2d(out_eigvec, in_eigvec, (size_t)F * K * D * D * 16, s)) != PBBSS_OK) return rc;
    if ((rc = copy_d2d(out_eigval, in_eigval, (size_t)F * K * D * 8, s)) != PBBSS_OK) return rc;
    if ((rc = copy_d2d(out_weight, in_weight, wcount * 8, s)) != PBBSS_OK) return rc;
    if ((rc = copy_d2d(out_mean, in_mean, (size_t)K * E * 8, s)) != PBBSS_OK) return rc;
    if ((rc = copy_d2d(out_scale, in_scale, nscale * 8, s)) != PBBSS_OK) return rc;
  }
  // spectral log-pdf (times spectral_weight) of every point, laid out (F,K,T)
  const bool fixed_scale = in_scale && has_gamma;
  bool mq_fresh = false;  // the full-covariance M-step also leaves the factorisation behind
  auto spectral = [&]() -> int {
    if (g_diag)
      return pbbss::launch_diag_estep(yd, o->embedding_is_f64, N, E, K, out_mean, out_scale,
                                      o->spectral_weight, T, dconst, slp, s);
    if (g_full) {
      int r = PBBSS_OK;
      if (!mq_fresh || fixed_scale)
        r = pbbss::launch_gauss_full_factor(out_scale, K, E, mq, offset, gst, s);
      if (r != PBBSS_OK) return r;
      r = pbbss::launch_gauss_full_logpdf(embedding, o->embedding_is_f64, 1, N, E, K, out_mean, mq,
                                          offset, nullptr, lpkn, nullptr, s);
      if (r != PBBSS_OK) return r;
      return pbbss::launch_kn_to_fkt(lpkn, o->spectral_weight, F, K, T, slp, s);
    }
    if (fixed_scale || o->iterations == 0) {  // otherwise the M-step finalize wrote them
      int r = pbbss::launch_embed_offsets(o->kind, K, E, out_scale, offset, prec, s);
      if (r != PBBSS_OK) return r;
    }
    return pbbss::launch_embed_estep(o->kind, yd, o->embedding_is_f64, 1, N, E, K, out_mean, prec,
                                     offset, nullptr, o->spectral_weight, T, slp, nullptr, s);
  };
  // generic-size spatial half: the posteriors of the current model (E-step of gcacgmm.py:66-117
  // with the spectral log-pdf as the extra exponent), then the cACG M-step from them
  const pbbss::GenInverseState g_state{g_inv, g_logdet, nullptr};
  auto gen_m_step = [&](const double* gam, bool fk_weights) -> int {
    int r = pbbss::launch_gen_mstep_cov(observation, o->obs_is_c128, F, T, D, K, g_mw, gam, saliency,
                                        PBBSS_WEIGHT_PER_CLASS_MEAN, g_csum, g_cov,
                                        fk_weights ? out_weight : tmp, s);
    if (r != PBBSS_OK) return r;
    return pbbss::launch_gen_heev(g_cov, (int64_t)nmat, D, o->covariance_norm, o->eigenvalue_floor,
                                  out_eigval, static_cast<double*>(out_eigvec), out_status,
                                  h->cfg.lds_limit, s);
  };
  auto gen_e_step = [&](double* aff_out, double eps, bool for_m_step, int inline_pa) -> int {
    if (inline_pa) {
      // spatial log-pdf and quadratic forms first, then the per-bin permutation search
      int r = pbbss::launch_gen_estep(g_yt, o->obs_is_c128, PBBSS_LAYOUT_DT, F, T, D, K,
                                      static_cast<const double*>(out_eigvec), out_eigval, out_weight,
                                      wb, wk, wt, nullptr, 0.0, nullptr, g_q, g_lp, s, g_state,
                                      nullptr, nullptr, nullptr, /*raw_dt=*/1);
      if (r != PBBSS_OK) return r;
      return pbbss::launch_gen_joint_pa(g_yt, o->obs_is_c128, F, T, D, K, g_lp, g_q, slp,
                                        o->spatial_weight, out_weight, wb, wk, wt, saliency, eps,
                                        aff_out, for_m_step ? g_mw : nullptr,
                                        for_m_step ? g_zero : nullptr, s);
    }
    return pbbss::launch_gen_estep(g_yt, o->obs_is_c128, PBBSS_LAYOUT_DT, F, T, D, K,
                                   static_cast<const double*>(out_eigvec), out_eigval, out_weight,
                                   wb, wk, wt, nullptr, eps, aff_out, nullptr, nullptr, s, g_state,
                                   saliency, for_m_step ? g_mw : nullptr,
                                   for_m_step ? g_zero : nullptr, /*raw_dt=*/1, slp,
                                   o->spatial_weight);
  };
  if (gen) {
    if ((rc = pbbss::launch_gen_transpose(observation, o->obs_is_c128, F, T, D, g_yt, s)) != PBBSS_OK)
      return rc;
    if (hipMemsetAsync(out_status, 0, nmat * sizeof(int32_t), s) != hipSuccess) return PBBSS_ERR_HIP;
    if (hipMemsetAsync(g_zero, 0, (size_t)F * sizeof(int32_t), s) != hipSuccess) return PBBSS_ERR_HIP;
  }
  auto joint = [&](int iterations, double* aff_out, int inline_pa, const double* state_in,
                   double* state_out, int emit_model) -> int {
    if (gen) {
      int r = gen_e_step(aff_out, iterations > 0 ? o->affiliation_eps : 0.0, iterations > 0,
                         inline_pa);
      if (r != PBBSS_OK || iterations == 0) return r;
      return gen_m_step(aff_out, o->weight_mode == PBBSS_JOINT_WEIGHT_FK);
    }
    pbbss::EmArgs a{};
    a.y = observation;
    a.B = F;
    a.T = T;
    a.in_eigvec = static_cast<const double*>(out_eigvec);  // in place: one workgroup per bin
    a.in_eigval = out_eigval;
    a.in_weight = out_weight;
    a.wb = wb;
    a.wk = wk;
    a.wt = wt;
    a.saliency = saliency;
    a.out_eigvec = static_cast<double*>(out_eigvec);
    a.out_eigval = out_eigval;
    a.out_status = out_status;
    a.out_aff = aff_out;
    a.iterations = iterations;
    a.covariance_norm = o->covariance_norm;
    a.weight_mode = PBBSS_WEIGHT_PER_CLASS_MEAN;
    a.layout = PBBSS_LAYOUT_TD;
    a.aff_eps = o->affiliation_eps;
    a.final_eps = 0.0;
    a.eig_floor = o->eigenvalue_floor;
    pbbss::JointExtras jx{slp, o->spatial_weight, nullptr, state_in, state_out, emit_model,
                          (iterations > 0 && o->weight_mode == PBBSS_JOINT_WEIGHT_FK) ? out_weight
                                                                                     : nullptr};
    return pbbss::joint_launch(D, K, o->obs_is_c128, a, jx, inline_pa, h->cfg, s);
  };
  // spatial half of the rotated loop: mode 0 = quadratic forms of the eigen model in the output
  // arrays, 1 = M-step from (G = aff, Q = slp) + factorisation + quadratic forms of the new
  // model (Q in place), 2 = M-step + exact eigen path, (V, lambda) emitted (last iteration)
  // in-launch finalize: not for sharded fits (the all-reduce of the partials has to sit between
  // the sweep and the finalize, in stream order); PBBSS_JOINT_INLAUNCH_FINALIZE=0 for A/B runs
  static const bool fin_allowed = [] {
    const char* v = getenv("PBBSS_JOINT_INLAUNCH_FINALIZE");
    return !(v && v[0] == '0');
  }();
  const bool fin_in_launch = rot && fin_allowed && !reduce && h->cfg.xbuf &&
                             2 * K * (E + 1) <= pbbss::kSpectralFinMaxW2;
  auto spatial_ms = [&](int mode) -> int {
    pbbss::EmArgs a{};
    a.y = observation;
    a.B = F;
    a.T = T;
    a.gamma0 = mode == 0 ? nullptr : aff;
    a.q0 = mode == 0 ? nullptr : slp;
    a.saliency = saliency;
    a.in_eigvec = static_cast<const double*>(out_eigvec);
    a.in_eigval = out_eigval;
    a.out_eigvec = static_cast<double*>(out_eigvec);
    a.out_eigval = out_eigval;
    a.out_status = out_status;
    a.iterations = 1;
    a.covariance_norm = o->covariance_norm;
    a.weight_mode = PBBSS_WEIGHT_PER_CLASS_MEAN;
    a.layout = PBBSS_LAYOUT_TD;
    a.eig_floor = o->eigenvalue_floor;
    pbbss::JointMs jm{};
    jm.mode = mode;
    jm.q_out = slp;
    jm.lndet_out = lndet;
    jm.weight_fk_out = (mode != 0 && o->weight_mode == PBBSS_JOINT_WEIGHT_FK) ? out_weight : nullptr;
    jm.fin.kind = -1;
    if (mode != 0 && fin_in_launch) {
      int C = 0;
      pbbss::joint_sweep_chunks(o->kind, N, E, K, o->embedding_is_f64, &C);
      jm.fin = pbbss::SpectralFin{o->kind == PBBSS_EMBED_VMF ? 0 : 1, joint_fin_helpers(), C, E, K,
                                  part, fin_tmp,
                                  reinterpret_cast<unsigned*>(h->cfg.xbuf + 224),  // free word
                                  o->min_concentration, o->max_concentration, out_mean, out_scale,
                                  offset, prec};
      if (jm.fin.helpers > C) jm.fin.helpers = C;
    }
    return pbbss::joint_ms_launch(D, K, o->obs_is_c128, a, jm, h->cfg, s);
  };
  for (int it = 0; it < o->iterations; ++it) {
    const double* src = gamma0;
    if (it == 0 && gen) {
      rc = pbbss::launch_gen_init_weights(g_yt, o->obs_is_c128, PBBSS_LAYOUT_DT, F, T, D, K, gamma0,
                                          saliency, g_mw, g_zero, s);
      if (rc != PBBSS_OK) return rc;
      if ((rc = gen_m_step(gamma0, false)) != PBBSS_OK) return rc;
    } else if (it == 0) {
      // first M-step from the initial affiliations, quadratic form = 1 (gcacgmm.py:194-196)
      pbbss::EmArgs a{};
      a.y = observation;
      a.B = F;
      a.T = T;
      a.gamma0 = gamma0;
      a.saliency = saliency;
      a.out_eigvec = static_cast<double*>(out_eigvec);
      a.out_eigval = out_eigval;
      a.out_status = out_status;
      a.iterations = 1;
      a.covariance_norm = o->covariance_norm;
      a.weight_mode = PBBSS_WEIGHT_PER_CLASS_MEAN;
      a.layout = PBBSS_LAYOUT_TD;
      a.eig_floor = o->eigenvalue_floor;
      rc = pbbss::em_launch(D, K, o->obs_is_c128, a, h->cfg, s);
      if (rc != PBBSS_OK) return rc;
    } else if (rot) {
      // sweep: posteriors of the current model -> aff, spectral sums -> part; then the spectral
      // finalize and the spatial M-step / factorisation / next quadratic forms
      rc = pbbss::launch_joint_sweep(o->kind, embedding, o->embedding_is_f64, F, T, E, K, D, slp,
                                     lndet, out_weight, wb, wk, wt, out_mean, prec, offset, o->spatial_weight,
                                     o->spectral_weight, saliency, o->affiliation_eps, aff, part, s);
      if (rc != PBBSS_OK) return rc;
      // The spectral finalize (ONE workgroup walking the chunk partials: 14-18 us) and the spatial
      // kernel are independent -- both only feed the NEXT sweep -- so the finalize runs beside
      // the spatial kernel on the handle's side stream (fork after the sweep, join before the
      // next sweep).  Sharded fits keep it in stream order (the all-reduce of the partials is
      // enqueued on the caller's stream).  PBBSS_JOINT_SIDE_FINALIZE=0: in stream order (A/B).
      static const bool side_allowed = [] {
        const char* v = getenv("PBBSS_JOINT_SIDE_FINALIZE");
        return !(v && v[0] == '0');
      }();
      const bool side = side_allowed && !reduce && h->cfg.side_stream && !fin_in_launch;
      hipStream_t fs = s;
      if (side) {
        if (hipEventRecord(h->cfg.ev_fork, s) != hipSuccess ||
            hipStreamWaitEvent(h->cfg.side_stream, h->cfg.ev_fork, 0) != hipSuccess)
          return PBBSS_ERR_HIP;
        fs = h->cfg.side_stream;
      }
      if (!fin_in_launch) {
        rc = pbbss::launch_joint_sweep_finalize(o->kind, embedding, o->embedding_is_f64, N, E, K,
                                                o->min_concentration, o->max_concentration, part,
                                                out_mean, out_scale, offset, prec, fs, reduce);
        if (rc != PBBSS_OK) return rc;
      }
      if (side && hipEventRecord(h->cfg.ev_join, fs) != hipSuccess) return PBBSS_ERR_HIP;
      if ((rc = spatial_ms(it == o->iterations - 1 ? 2 : 1)) != PBBSS_OK) return rc;
      if (o->weight_mode != PBBSS_JOINT_WEIGHT_FK) {
        rc = pbbss::launch_joint_weight(o->weight_mode, aff, saliency, F, K, T, tmp, out_weight, s,
                                        reduce);
        if (rc != PBBSS_OK) return rc;
      }
      if (side && hipStreamWaitEvent(s, h->cfg.ev_join, 0) != hipSuccess) return PBBSS_ERR_HIP;
      continue;
    } else {
      if ((rc = spectral()) != PBBSS_OK) return rc;
      // the model travels as packed inverse covariances between iterations; the first joint
      // step reads the eigen model of the initial M-step, the last one emits (V, lambda)
      const bool last = (it == o->iterations - 1);
      rc = joint(1, aff, o->inline_pa, it == 1 ? nullptr : jstate, last ? nullptr : jstate,
                 last ? 1 : 0);
      if (rc != PBBSS_OK) return rc;
      src = aff;
    }
    if (it == 0 || o->weight_mode != PBBSS_JOINT_WEIGHT_FK) {  // 'fk' weights: joint kernel
      rc = pbbss::launch_joint_weight(o->weight_mode, src, saliency, F, K, T, tmp, out_weight, s,
                                      reduce);
      if (rc != PBBSS_OK) return rc;
    }
    if (g_full) {
      // GaussianTrainer._fit(covariance_type='full') on the (1, F*T, E) embedding with the masked
      // affiliations as (K, F*T) class weights (gcacgmm.py:297-307); the kernel leaves mean,
      // covariance and the factorisation the next E-step needs
      if ((rc = pbbss::launch_fkt_to_kn(src, saliency, F, K, T, wkn, s)) != PBBSS_OK) return rc;
      rc = pbbss::launch_gauss_full_fit(embedding, o->embedding_is_f64, 1, N, E, K, wkn, nullptr,
                                        gpart, out_mean, out_scale, mq, offset, nullptr, gst, s,
                                        sharded ? gshift : nullptr, reduce);
      mq_fresh = true;
    } else {
      rc = pbbss::launch_embed_fit(o->kind, embedding, o->embedding_is_f64, 1, N, E, K, src, T,
                                   saliency, o->min_concentration, o->max_concentration, -1, part,
                                   out_mean, out_scale, nullptr, g_diag ? nullptr : offset,
                                   g_diag ? nullptr : prec, it == 0 ? 2 : 1, s, nullptr, reduce);
    }
    if (rc != PBBSS_OK) return rc;
    if (fixed_scale) {  // fixed_covariance (gcacgmm.py:305-312)
      if ((rc = copy_d2d(out_scale, in_scale, nscale * 8, s)) != PBBSS_OK) return rc;
    }
    if (rot && it == 0) {  // quadratic forms of the first model for the first sweep
      if ((rc = spatial_ms(0)) != PBBSS_OK) return rc;
    }
  }
  if (o->final_predict && out_affiliation) {
    if ((rc = spectral()) != PBBSS_OK) return rc;
    if ((rc = joint(0, out_affiliation, 0, nullptr, nullptr, 0)) != PBBSS_OK) return rc;
  }
  // a spectral covariance that stopped being positive definite (the reference raises from
  // sklearn's precision Cholesky, gaussian.py:26): PBBSS_ST_NOT_POSDEF in status word 0
  if (g_full) return pbbss::launch_or_status(gst, out_status, s);
  return PBBSS_OK;
}

// ---------------------------------------------------------------------------
// N4: remaining beamformer family (bf_extra.hip)
// ---------------------------------------------------------------------------
PBBSS_API int pbbss_lcmv(pbbss_handle_t h, const void* atf, const void* response,
                         const void* noise, int64_t F, int D, int K, void* out_w,
                         int32_t* out_status, void* stream) {
  DeviceGuard device_guard(h);
  if (!h || !atf || !response || !noise || !out_w || F <= 0) return PBBSS_ERR_INVALID_ARG;
  return pbbss::launch_lcmv(static_cast<const double*>(atf), static_cast<const double*>(response),
                            static_cast<const double*>(noise), F, D, K,
                            static_cast<double*>(out_w), out_status, as_stream(stream));
}

PBBSS_API int pbbss_phase_correction(pbbss_handle_t h, const void* vector, int64_t lead,
                                     int64_t rest, int F, int D, int two_d, void* scratch,
                                     void* out, void* stream) {
  DeviceGuard device_guard(h);
  if (!h || !vector || !out || lead <= 0 || rest <= 0 || F <= 0 || D <= 0)
    return PBBSS_ERR_INVALID_ARG;
  if (F > 1 && !scratch) return PBBSS_ERR_INVALID_ARG;
  if (two_d && (lead != 1 || rest != 1)) return PBBSS_ERR_INVALID_ARG;
  return pbbss::launch_phase_correction(static_cast<const double*>(vector), lead, rest, F, D,
                                        two_d, static_cast<double*>(scratch),
                                        static_cast<double*>(out), as_stream(stream));
}

PBBSS_API int pbbss_snr_postfilter(pbbss_handle_t h, const void* w, const void* target,
                                   const void* noise, int64_t F, int D, void* out, void* stream) {
  DeviceGuard device_guard(h);
  if (!h || !w || !target || !noise || !out || F <= 0 || D <= 0) return PBBSS_ERR_INVALID_ARG;
  return pbbss::launch_bf_quadratic(0, static_cast<const double*>(w),
                                    static_cast<const double*>(target),
                                    static_cast<const double*>(noise), nullptr, F, D,
                                    static_cast<double*>(out), as_stream(stream));
}

PBBSS_API int pbbss_reference_channel_terms(pbbss_handle_t h, const void* w_mat, const void* target,
                                            const void* noise, int64_t F, int D, void* out_num,
                                            void* out_den, void* stream) {
  DeviceGuard device_guard(h);
  if (!h || !w_mat || !target || !noise || !out_num || !out_den || F <= 0 || D <= 0)
    return PBBSS_ERR_INVALID_ARG;
  return pbbss::launch_refch_terms(static_cast<const double*>(w_mat),
                                   static_cast<const double*>(target),
                                   static_cast<const double*>(noise), F, D,
                                   static_cast<double*>(out_num), static_cast<double*>(out_den),
                                   as_stream(stream));
}

PBBSS_API int pbbss_rank_one_approximation(pbbss_handle_t h, const void* covariance,
                                           const void* vector, int64_t N, int D, void* out,
                                           void* stream) {
  DeviceGuard device_guard(h);
  if (!h || !covariance || !vector || !out || N <= 0 || D <= 0) return PBBSS_ERR_INVALID_ARG;
  return pbbss::launch_rank_one(static_cast<const double*>(covariance),
                                static_cast<const double*>(vector), N, D,
                                static_cast<double*>(out), as_stream(stream));
}

PBBSS_API int pbbss_matvec(pbbss_handle_t h, const void* matrix, const void* vector, int64_t N,
                           int D, void* out, void* stream) {
  DeviceGuard device_guard(h);
  if (!h || !matrix || !vector || !out || N <= 0 || D <= 0) return PBBSS_ERR_INVALID_ARG;
  return pbbss::launch_matvec(static_cast<const double*>(matrix),
                              static_cast<const double*>(vector), N, D, static_cast<double*>(out),
                              as_stream(stream));
}

PBBSS_API int pbbss_distortionless_normalization(pbbss_handle_t h, const void* w, const void* atf,
                                                 const void* noise, int64_t F, int D, void* out,
                                                 void* stream) {
  DeviceGuard device_guard(h);
  if (!h || !w || !atf || !noise || !out || F <= 0 || D <= 0) return PBBSS_ERR_INVALID_ARG;
  return pbbss::launch_bf_quadratic(1, static_cast<const double*>(w), nullptr,
                                    static_cast<const double*>(noise),
                                    static_cast<const double*>(atf), F, D,
                                    static_cast<double*>(out), as_stream(stream));
}

PBBSS_API int pbbss_zero_degree_normalization(pbbss_handle_t h, const void* vector, int64_t N,
                                              int D, int reference_channel, void* out,
                                              void* stream) {
  DeviceGuard device_guard(h);
  if (!h || !vector || !out || N <= 0 || D <= 0) return PBBSS_ERR_INVALID_ARG;
  if (reference_channel < 0 || reference_channel >= D) return PBBSS_ERR_INVALID_ARG;
  return pbbss::launch_zero_degree(static_cast<const double*>(vector), N, D, reference_channel,
                                   static_cast<double*>(out), as_stream(stream));
}

PBBSS_API int pbbss_condition_covariance(pbbss_handle_t h, const void* x, int64_t N, int D,
                                         double gamma, void* out, void* stream) {
  DeviceGuard device_guard(h);
  if (!h || !x || !out || N <= 0 || D <= 0) return PBBSS_ERR_INVALID_ARG;
  // one thread per entry: the diagonal threads read the whole diagonal of x while their
  // neighbours store to out -- in place the trace would pick up conditioned entries
  if (x == out) return PBBSS_ERR_INVALID_ARG;
  return pbbss::launch_condition_covariance(static_cast<const double*>(x), N, D, gamma,
                                            static_cast<double*>(out), as_stream(stream));
}

PBBSS_API int pbbss_apply_online_beamforming_vector(pbbss_handle_t h, const void* vector,
                                                    const void* mix, int mix_is_c128, int64_t F,
                                                    int T, int D, void* out, void* stream) {
  DeviceGuard device_guard(h);
  if (!h || !vector || !mix || !out || F <= 0 || T <= 0 || D <= 0) return PBBSS_ERR_INVALID_ARG;
  return pbbss::launch_apply_online(static_cast<const double*>(vector), mix, mix_is_c128, F, T, D,
                                    static_cast<double*>(out), as_stream(stream));
}

// ---------------------------------------------------------------- STFT edge (row N4)
PBBSS_API int pbbss_stft_num_frames(int64_t num_samples, int size, int shift, int window_length,
                                    int fading, int pad) {
  if (num_samples < 0 || size < 1 || shift < 1 || window_length < 1) return PBBSS_ERR_INVALID_ARG;
  const int64_t n = num_samples + (fading ? 2 * (int64_t)(window_length - shift) : 0);
  if (n < window_length) return pad ? 1 : 0;
  const int64_t rest = n - window_length;
  const int64_t frames = 1 + (pad ? (rest + shift - 1) / shift : rest / shift);
  return frames > INT32_MAX ? PBBSS_ERR_UNSUPPORTED : (int)frames;
}

PBBSS_API int pbbss_stft(pbbss_handle_t h, const void* x, int x_is_f64, int64_t C, int64_t N,
                         int size, int shift, int window_length, const double* window, int fading,
                         int pad, int out_layout, int out_is_c128, void* out, void* stream) {
  DeviceGuard device_guard(h);
  if (!h || !x || !window || !out || C <= 0 || N <= 0) return PBBSS_ERR_INVALID_ARG;
  if (fading && window_length < shift) return PBBSS_ERR_INVALID_ARG;
  const int T = pbbss_stft_num_frames(N, size, shift, window_length, fading, pad);
  if (T <= 0) return T < 0 ? T : PBBSS_ERR_INVALID_ARG;
  return pbbss::launch_stft(x, x_is_f64, C, N, size, shift, window_length, window,
                            fading ? window_length - shift : 0, T, out_layout, out_is_c128, out,
                            h->cfg.lds_limit, as_stream(stream));
}

PBBSS_API int pbbss_istft(pbbss_handle_t h, const void* X, int x_is_c128, int64_t C, int T, int size,
                          int shift, int window_length, const double* synthesis_window, int fading,
                          double* out, int64_t n_out, void* stream) {
  DeviceGuard device_guard(h);
  if (!h || !X || !synthesis_window || !out || C <= 0 || T <= 0) return PBBSS_ERR_INVALID_ARG;
  if (window_length < shift || shift < 1) return PBBSS_ERR_INVALID_ARG;
  const int fade = fading ? window_length - shift : 0;
  if (n_out != (int64_t)T * shift + window_length - shift - 2 * (int64_t)fade) return PBBSS_ERR_INVALID_ARG;
  void* wmem = handle_work(h, WorkCarver::pad((size_t)C * T * window_length * 8));
  if (!wmem) return PBBSS_ERR_HIP;
  return pbbss::launch_istft(X, x_is_c128, C, T, size, shift, window_length, synthesis_window, fade,
                             static_cast<double*>(wmem), out, n_out, h->cfg.lds_limit,
                             as_stream(stream));
}
