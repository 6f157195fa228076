// allreduce.hip -- tensor-parallel all-reduce over RCCL/xGMI (include/dashinfer_hip.h section 6).
//
// Replaces AllReduceOp's ncclAllReduce + host Synchronize
// (csrc/core/operator/nccl/allreduce/allreduce_op.cpp:84-92).  The collective is enqueued on the
// caller's stream and never blocks the host, so the caller can run it on a side stream and
// overlap it with independent work.
#include <rccl/rccl.h>

#include <cstring>

#include "device_utils.h"

using namespace dihip;

static_assert(sizeof(ncclUniqueId) == 128, "dihip_rccl_unique_id exchanges 128 bytes");

extern "C" {

int dihip_rccl_unique_id(void* id128_host) {
  DIHIP_REQUIRE(id128_host, DIHIP_PARAM_ERROR, "rccl_unique_id: null pointer");
  ncclUniqueId id;
  ncclResult_t r = ncclGetUniqueId(&id);
  DIHIP_REQUIRE(r == ncclSuccess, DIHIP_RUNTIME_ERROR, "ncclGetUniqueId: %s", ncclGetErrorString(r));
  std::memcpy(id128_host, &id, sizeof(id));
  return DIHIP_SUCCESS;
}

int dihip_rccl_comm_init_rank(void** comm, int nranks, const void* id128_host, int rank) {
  DIHIP_REQUIRE(comm && id128_host && nranks > 0 && rank >= 0 && rank < nranks, DIHIP_PARAM_ERROR,
                "rccl_comm_init_rank: bad argument");
  ncclUniqueId id;
  std::memcpy(&id, id128_host, sizeof(id));
  ncclComm_t c = nullptr;
  ncclResult_t r = ncclCommInitRank(&c, nranks, id, rank);
  DIHIP_REQUIRE(r == ncclSuccess, DIHIP_RUNTIME_ERROR, "ncclCommInitRank: %s", ncclGetErrorString(r));
  *comm = c;
  return DIHIP_SUCCESS;
}

int dihip_rccl_comm_destroy(void* comm) {
  if (!comm) return DIHIP_SUCCESS;
  ncclResult_t r = ncclCommDestroy(reinterpret_cast<ncclComm_t>(comm));
  DIHIP_REQUIRE(r == ncclSuccess, DIHIP_RUNTIME_ERROR, "ncclCommDestroy: %s", ncclGetErrorString(r));
  return DIHIP_SUCCESS;
}

int dihip_allreduce_sum(void* comm, void* stream, const void* in, void* out, size_t count, int dtype) {
  DIHIP_REQUIRE(comm && in && out, DIHIP_PARAM_ERROR, "allreduce: null pointer");
  if (count == 0) return DIHIP_SUCCESS;
  ncclDataType_t t;
  switch (dtype) {  // GetNcclType, csrc/device/cuda/nccl_utils.hpp:9-27
    case DIHIP_F32: t = ncclFloat32; break;
    case DIHIP_F16: t = ncclFloat16; break;
    case DIHIP_BF16: t = ncclBfloat16; break;
    default: set_last_error("allreduce: unsupported dtype %d", dtype); return DIHIP_PARAM_ERROR;
  }
  ncclResult_t r = ncclAllReduce(in, out, count, t, ncclSum, reinterpret_cast<ncclComm_t>(comm),
                                 reinterpret_cast<hipStream_t>(stream));
  DIHIP_REQUIRE(r == ncclSuccess, DIHIP_RUNTIME_ERROR, "ncclAllReduce: %s", ncclGetErrorString(r));
  return DIHIP_SUCCESS;
}

int dihip_allgather_bytes(void* comm, void* stream, const void* in, void* out, size_t bytes_per_rank) {
  DIHIP_REQUIRE(comm && in && out, DIHIP_PARAM_ERROR, "allgather: null pointer");
  if (bytes_per_rank == 0) return DIHIP_SUCCESS;
  ncclResult_t r = ncclAllGather(in, out, bytes_per_rank, ncclUint8, reinterpret_cast<ncclComm_t>(comm),
                                 reinterpret_cast<hipStream_t>(stream));
  DIHIP_REQUIRE(r == ncclSuccess, DIHIP_RUNTIME_ERROR, "ncclAllGather: %s", ncclGetErrorString(r));
  return DIHIP_SUCCESS;
}

}  // extern "C"
