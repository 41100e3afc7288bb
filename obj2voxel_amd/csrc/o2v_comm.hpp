// o2v_comm.hpp -- the few collectives the sharded (multi-GPU) voxelization needs, behind one small interface.
//
// the grid is split into z-slabs, triangles are duplicated into every slab they overlap and the
// voxel walk is clamped to the slab - the reference's own chunk mechanism (src/obj2voxel.cpp:226-243,
// src/voxelization.cpp:440-444) with one "chunk" per GPU - so voxel data never crosses GPUs.  What does cross them is
// planning data: the mesh bounds (6 floats, min / max), the z histogram of predicted work (2048 x u64, sum), the z extent
// of every block of 256 triangles (8 bytes per block, gathered) and the per-slab voxel counts (one u64 per rank, gathered)
// - the passes over the triangle list that produce them are sharded over the ranks instead of replicated.
//
// Implementations (o2v_comm.cpp):
//   RcclComm      RCCL over xGMI, on the caller's HIP stream, device buffers in place.  librccl is loaded with dlopen at
//                 first use (no link-time dependency: a process that already carries an RCCL, e.g. torch's, shares it).
//   CallbackComm  host-memory callbacks supplied by the embedding program (a torch.distributed gloo group in the CPU-side
//                 tests; the shared-memory exchange between the threads of an in-process group when RCCL cannot be used,
//                 e.g. two ranks on one GPU).  The device buffers are staged through host memory.
#pragma once

#include "../../include/o2v_hip.h"

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <string>

struct o2v_hip_comm {
    int rank = 0, world = 1;
    std::string err;
    // A collective of this communicator is stuck on a stream (a rank never arrived and the time limit passed): it must not be
    // waited for again - the destructor then aborts the communicator (ncclCommAbort) or leaves it to the process' end.
    bool poisoned = false;
    virtual ~o2v_hip_comm() {}
    virtual const char *kind() const = 0;
    // All buffers are device memory of the calling rank; operations are in place and ordered on `stream`.
    // Element-wise min over ranks of n uint32 values.
    virtual int allreduce_min_u32(uint32_t *d_buf, size_t n, hipStream_t stream) = 0;
    virtual int allreduce_max_u32(uint32_t *d_buf, size_t n, hipStream_t stream) = 0;
    // Element-wise sum over ranks of n uint64 values.
    virtual int allreduce_sum_u64(unsigned long long *d_buf, size_t n, hipStream_t stream) = 0;
    // d_buf holds world * bytes_per_rank bytes; rank r's part sits at d_buf + r * bytes_per_rank and is sent to everyone.
    virtual int allgather(void *d_buf, size_t bytes_per_rank, hipStream_t stream) = 0;
    // root's d_buf[0, bytes) is copied to every rank's d_buf.
    virtual int broadcast(void *d_buf, size_t bytes, int root, hipStream_t stream) = 0;
};

namespace o2v {

// nullptr + error text in `err` if RCCL is not available or the communicator cannot be created.
o2v_hip_comm *make_rccl_comm(const uint8_t id[O2V_HIP_COMM_ID_BYTES], int rank, int world, int device, std::string &err);
bool rccl_unique_id(uint8_t id[O2V_HIP_COMM_ID_BYTES], std::string &err);
o2v_hip_comm *make_callback_comm(const o2v_hip_comm_callbacks &cb, int rank, int world);
// What a rank waits for another rank at most (communicator creation, the readiness all-reduce of a sharded run) before it
// fails with a message instead of hanging: O2V_COMM_TIMEOUT_S seconds, 120 by default.
double comm_timeout_seconds();
bool stream_wait_limited(hipStream_t s, const char *what, std::string &err);

}  // namespace o2v
