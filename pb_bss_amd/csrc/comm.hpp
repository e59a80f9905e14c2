// RCCL communicator + mask all-gather behind the C ABI (comm.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace pbbss {
int comm_unique_id(void* out_id_128_bytes);
int comm_create(const void* id, int world, int rank, void** out_comm);
int comm_destroy(void* comm);
int comm_all_gather_bytes(void* comm, const void* send, void* recv, size_t bytes_per_rank,
                          hipStream_t s);
int comm_all_reduce_f64(void* comm, double* buf, size_t count, hipStream_t s);
int comm_query(void* comm, int* world, int* rank);
// local (outer, nloc, inner) -> (outer, pad, inner), zero rows appended
int launch_allgather_pack(const void* local, int elem_bytes, int64_t outer, int64_t nloc,
                          int64_t pad, int64_t inner, void* out, hipStream_t s);
// gathered (world, outer, pad, inner), pad = ceil(total_bins / world) -> (outer, total_bins, inner)
int launch_allgather_unpack(const void* gathered, int elem_bytes, int world, int64_t outer,
                            int64_t total_bins, int64_t inner, void* out, hipStream_t s);
}  // namespace pbbss
