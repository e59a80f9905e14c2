// The one real exchange step of the path behind the C ABI: the all-gather of the posterior masks
// over RCCL / xGMI (SURVEY.md section 8b `pbbss_allgather_masks`, 8e) -- so that a host that is
// not Python (no torch.distributed) can shard the frequency bins over the GPUs of a node.
//
// One process per GPU.  The communicator lives in the handle: rank 0 obtains a 128-byte RCCL
// unique id (pbbss_comm_unique_id), the host distributes it by whatever means it has, every
// rank calls pbbss_comm_create.  RCCL is resolved at run time (dlopen "librccl.so.1"): the
// library keeps loading on boxes without RCCL, and a process that already carries torch's copy
// of RCCL shares it instead of pulling in a second one.
//
// Shards are contiguous blocks of bins whose sizes differ by at most one (513 = 65 + 7 x 64):
// every rank pads its block to the largest one (pack kernel), ONE ncclAllGather moves
// world x outer x pad x inner elements, and an unpack kernel trims the padding while writing
// the (outer, total_bins, inner) result.  xGMI is a full mesh, the message is a few hundred KB
// to a few hundred MB: a single collective, no bucketing.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "comm.hpp"
#include "pbbss.h"

namespace pbbss {
namespace {

// the slice of the RCCL API this file uses (rccl.h: ncclGetUniqueId :187, ncclCommInitRank :220,
// ncclCommDestroy :260, ncclAllGather :678; ncclUint8 = 1)
struct UniqueId {
  char internal[128];
};
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(void**, int, UniqueId, int);
typedef int (*CommDestroyFn)(void*);
typedef int (*AllGatherFn)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, void*, hipStream_t);  // :611
typedef int (*CommCountFn)(void*, int*);                                               // :378

struct Rccl {
  void* lib = nullptr;
  GetUniqueIdFn get_unique_id = nullptr;
  CommInitRankFn comm_init_rank = nullptr;
  CommDestroyFn comm_destroy = nullptr;
  AllGatherFn all_gather = nullptr;
  AllReduceFn all_reduce = nullptr;
  CommCountFn comm_count = nullptr, comm_user_rank = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r = [] {
    Rccl x;
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      x.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (x.lib) break;
    }
    if (!x.lib) return x;
    x.get_unique_id = reinterpret_cast<GetUniqueIdFn>(dlsym(x.lib, "ncclGetUniqueId"));
    x.comm_init_rank = reinterpret_cast<CommInitRankFn>(dlsym(x.lib, "ncclCommInitRank"));
    x.comm_destroy = reinterpret_cast<CommDestroyFn>(dlsym(x.lib, "ncclCommDestroy"));
    x.all_gather = reinterpret_cast<AllGatherFn>(dlsym(x.lib, "ncclAllGather"));
    x.all_reduce = reinterpret_cast<AllReduceFn>(dlsym(x.lib, "ncclAllReduce"));
    x.comm_count = reinterpret_cast<CommCountFn>(dlsym(x.lib, "ncclCommCount"));
    x.comm_user_rank = reinterpret_cast<CommCountFn>(dlsym(x.lib, "ncclCommUserRank"));
    x.ok = x.get_unique_id && x.comm_init_rank && x.comm_destroy && x.all_gather && x.all_reduce;
    return x;
  }();
  return r;
}

// local (outer, nloc, inner) -> padded (outer, pad, inner), zeros in the padding rows
template <typename T>
__global__ void pack_kernel(const T* __restrict__ local, int64_t outer, int64_t nloc, int64_t pad,
                            int64_t inner, T* __restrict__ out) {
  const int64_t total = outer * pad * inner;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t in = i % inner, b = (i / inner) % pad, o = i / (inner * pad);
    out[i] = (b < nloc) ? local[(o * nloc + b) * inner + in] : T(0);
  }
}

// gathered (world, outer, pad, inner) -> out (outer, total_bins, inner); rank r owns the bins
// [start_r, start_r + size_r), size_r = base + (r < extra)
template <typename T>
__global__ void unpack_kernel(const T* __restrict__ gathered, int world, int64_t outer,
                              int64_t total_bins, int64_t pad, int64_t inner,
                              T* __restrict__ out) {
  const int64_t base = total_bins / world, extra = total_bins % world;
  const int64_t total = outer * total_bins * inner;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t in = i % inner, f = (i / inner) % total_bins, o = i / (inner * total_bins);
    // the first `extra` ranks hold base + 1 bins
    const int64_t split = extra * (base + 1);
    int64_t r, b;
    if (f < split) {
      r = f / (base + 1);
      b = f % (base + 1);
    } else {
      r = extra + (f - split) / (base > 0 ? base : 1);
      b = (f - split) % (base > 0 ? base : 1);
    }
    out[i] = gathered[((r * outer + o) * pad + b) * inner + in];
  }
}

unsigned grid_for(int64_t n) {
  int64_t g = (n + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

int comm_unique_id(void* out_id) {
  if (!rccl().ok) return PBBSS_ERR_UNSUPPORTED;
  return rccl().get_unique_id(static_cast<UniqueId*>(out_id)) == 0 ? PBBSS_OK : PBBSS_ERR_HIP;
}

int comm_create(const void* id, int world, int rank, void** out_comm) {
  if (!rccl().ok) return PBBSS_ERR_UNSUPPORTED;
  UniqueId uid;
  __builtin_memcpy(&uid, id, sizeof(uid));
  void* c = nullptr;
  if (rccl().comm_init_rank(&c, world, uid, rank) != 0) return PBBSS_ERR_HIP;
  *out_comm = c;
  return PBBSS_OK;
}

int comm_destroy(void* comm) {
  if (!comm) return PBBSS_OK;
  if (!rccl().ok) return PBBSS_ERR_UNSUPPORTED;
  return rccl().comm_destroy(comm) == 0 ? PBBSS_OK : PBBSS_ERR_HIP;
}

int launch_allgather_pack(const void* local, int elem_bytes, int64_t outer, int64_t nloc,
                          int64_t pad, int64_t inner, void* out, hipStream_t s) {
  const int64_t n = outer * pad * inner;
  if (n == 0) return PBBSS_OK;
  if (elem_bytes == 8) {
    hipLaunchKernelGGL(pack_kernel<double>, dim3(grid_for(n)), dim3(256), 0, s,
                       static_cast<const double*>(local), outer, nloc, pad, inner,
                       static_cast<double*>(out));
  } else {
    hipLaunchKernelGGL(pack_kernel<float>, dim3(grid_for(n)), dim3(256), 0, s,
                       static_cast<const float*>(local), outer, nloc, pad, inner,
                       static_cast<float*>(out));
  }
  return hipGetLastError() == hipSuccess ? PBBSS_OK : PBBSS_ERR_HIP;
}

int launch_allgather_unpack(const void* gathered, int elem_bytes, int world, int64_t outer,
                            int64_t total_bins, int64_t inner, void* out, hipStream_t s) {
  const int64_t n = outer * total_bins * inner;
  if (n == 0) return PBBSS_OK;
  const int64_t pad = (total_bins + world - 1) / world;
  if (elem_bytes == 8) {
    hipLaunchKernelGGL(unpack_kernel<double>, dim3(grid_for(n)), dim3(256), 0, s,
                       static_cast<const double*>(gathered), world, outer, total_bins, pad, inner,
                       static_cast<double*>(out));
  } else {
    hipLaunchKernelGGL(unpack_kernel<float>, dim3(grid_for(n)), dim3(256), 0, s,
                       static_cast<const float*>(gathered), world, outer, total_bins, pad, inner,
                       static_cast<float*>(out));
  }
  return hipGetLastError() == hipSuccess ? PBBSS_OK : PBBSS_ERR_HIP;
}

int comm_all_gather_bytes(void* comm, const void* send, void* recv, size_t bytes_per_rank,
                          hipStream_t s) {
  if (!rccl().ok || !comm) return PBBSS_ERR_UNSUPPORTED;
  return rccl().all_gather(send, recv, bytes_per_rank, /*ncclUint8*/ 1, comm, s) == 0
             ? PBBSS_OK
             : PBBSS_ERR_HIP;
}

// in-place sum of `count` float64 over the ranks (ncclFloat64 = 8, ncclSum = 0): every rank ends
// up with bit-identical values, which is what keeps the replicated spectral models of a sharded
// joint fit consistent
int comm_all_reduce_f64(void* comm, double* buf, size_t count, hipStream_t s) {
  if (!rccl().ok || !comm) return PBBSS_ERR_UNSUPPORTED;
  if (count == 0) return PBBSS_OK;
  return rccl().all_reduce(buf, buf, count, /*ncclFloat64*/ 8, /*ncclSum*/ 0, comm, s) == 0
             ? PBBSS_OK
             : PBBSS_ERR_HIP;
}

// what RCCL itself reports for the communicator (not what the caller passed to comm_create)
int comm_query(void* comm, int* world, int* rank) {
  if (!rccl().ok || !comm || !rccl().comm_count || !rccl().comm_user_rank)
    return PBBSS_ERR_UNSUPPORTED;
  if (rccl().comm_count(comm, world) != 0 || rccl().comm_user_rank(comm, rank) != 0)
    return PBBSS_ERR_HIP;
  return PBBSS_OK;
}

}  // namespace pbbss
