// allreduce.hip -- tensor-parallel all-reduce over RCCL/xGMI (include/dashinfer_hip.h section 6).
//
// Replaces AllReduceOp's ncclAllReduce + host Synchronize
// (csrc/core/operator/nccl/allreduce/allreduce_op.cpp:84-92).  The collective is enqueued on the
// caller's stream and never blocks the host, so the caller can run it on a side stream and
// overlap it with independent work.
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>

#include "device_utils.h"

using namespace dihip;

static_assert(sizeof(ncclUniqueId) == 128, "dihip_rccl_unique_id exchanges 128 bytes");

// rank-major [nranks][rows][row_bytes] -> row-major [rows][nranks * row_bytes] (transpose_axis_01 of the reference's
// nccl_allgather_launcher, allgather_op.cpp:27-58), 16 / 4 / 1 bytes per thread by alignment
template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(T* __restrict__ out, const T* __restrict__ tmp, int nranks, int rows,
                                                          size_t row_elems) {
  const size_t total = (size_t)nranks * rows * row_elems;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t e = i % row_elems, rr = i / row_elems;
    const size_t r = rr % nranks, m = rr / nranks;  // output element (row m, rank r, element e)
    out[i] = tmp[(r * rows + m) * row_elems + e];
  }
}

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

int dihip_gather_rows_transpose(void* stream, void* out, const void* tmp, int nranks, int rows, size_t row_bytes) {
  DIHIP_REQUIRE(out && tmp && nranks >= 1 && rows >= 0, DIHIP_PARAM_ERROR, "gather_rows_transpose: bad argument");
  if (rows == 0 || row_bytes == 0) return DIHIP_SUCCESS;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t total = (size_t)nranks * rows * row_bytes;
  const bool a16 = row_bytes % 16 == 0 && ((reinterpret_cast<uintptr_t>(tmp) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  const bool a4 = row_bytes % 4 == 0 && ((reinterpret_cast<uintptr_t>(tmp) | reinterpret_cast<uintptr_t>(out)) & 3) == 0;
  const size_t unit = a16 ? 16 : a4 ? 4 : 1;
  const int blocks = (int)std::max<size_t>(1, std::min<size_t>(1024, (total / unit + 255) / 256));
  if (a16)
    hipLaunchKernelGGL(gather_rows_kernel<u32x4_t>, dim3(blocks), dim3(256), 0, s, reinterpret_cast<u32x4_t*>(out),
                       reinterpret_cast<const u32x4_t*>(tmp), nranks, rows, row_bytes / 16);
  else if (a4)
    hipLaunchKernelGGL(gather_rows_kernel<uint32_t>, dim3(blocks), dim3(256), 0, s, reinterpret_cast<uint32_t*>(out),
                       reinterpret_cast<const uint32_t*>(tmp), nranks, rows, row_bytes / 4);
  else
    hipLaunchKernelGGL(gather_rows_kernel<unsigned char>, dim3(blocks), dim3(256), 0, s, reinterpret_cast<unsigned char*>(out),
                       reinterpret_cast<const unsigned char*>(tmp), nranks, rows, row_bytes);
  return launch_status();
}

int dihip_allgather_rows(void* comm, void* stream, const void* in, void* tmp, void* out, int rows, size_t row_bytes, int nranks) {
  DIHIP_REQUIRE(in && out && rows >= 0 && nranks >= 1, DIHIP_PARAM_ERROR, "allgather_rows: bad argument");
  if (rows == 0 || row_bytes == 0) return DIHIP_SUCCESS;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (nranks == 1) {
    if (in != out) {
      hipError_t e = hipMemcpyAsync(out, in, (size_t)rows * row_bytes, hipMemcpyDeviceToDevice, s);
      DIHIP_REQUIRE(e == hipSuccess, DIHIP_RUNTIME_ERROR, "allgather_rows: copy: %s", hipGetErrorString(e));
    }
    return DIHIP_SUCCESS;
  }
  DIHIP_REQUIRE(comm && tmp, DIHIP_PARAM_ERROR, "allgather_rows: communicator / workspace missing");
  ncclResult_t r = ncclAllGather(in, tmp, (size_t)rows * row_bytes, ncclUint8, reinterpret_cast<ncclComm_t>(comm), s);
  DIHIP_REQUIRE(r == ncclSuccess, DIHIP_RUNTIME_ERROR, "ncclAllGather: %s", ncclGetErrorString(r));
  return dihip_gather_rows_transpose(stream, out, tmp, nranks, rows, row_bytes);
}

}  // extern "C"
