// oracle/logits_ref.hip -- TEST INFRASTRUCTURE ONLY (never linked into, loaded by or shipped with the product).
//
// Runs the REFERENCE'S OWN logits processors -- the kernels rep_logits_processor / token_count_processor / penalty_logits_processor /
// n_gram_logits_processor / min_length_logits_processor and their launcher cuda::LogitsProcessor<T>, csrc/core/kernel/cuda/beam_search.cu:
// 329-539, and the BatchGencfg struct of csrc/common/common.h:271-282 -- compiled for gfx950 from slices of those two files taken where they
// lie (oracle/Makefile, target reflogits: awk writes the slices into oracle/_ref/ as build intermediates and removes them after the compile).
// What stands in for CUDA here is the runtime spelling only: cudaStream_t / cudaMemcpyAsync / cudaMemsetAsync are the HIP calls of the same
// meaning, AS_CHECK_CUDA evaluates its argument, THREAD_PER_BLOCK is the reference's 256 (cuda_common.h:41).  The kernels' bodies, their launch
// geometry, the copy of the scores and the memset of the count array are the reference's.
//
// Built with -ffp-contract=off: `count * frequency` and `+ presence` are two roundings, as in the product and in oracle/logits_proc.py.  (A
// -ffp-contract=fast build of the same slice gave identical bits on the test inputs -- hipcc emits no fused form for the conditional add; what
// nvcc's default -fmad=true picks on the reference's own platform cannot be reproduced here: at most one rounding of the penalty.)
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

#define THREAD_PER_BLOCK 256
#define AS_CHECK_CUDA(x) (void)(x)
#define cudaStream_t hipStream_t
#define cudaMemcpyAsync hipMemcpyAsync
#define cudaMemsetAsync hipMemsetAsync
#define cudaMemcpyDeviceToDevice hipMemcpyDeviceToDevice

namespace allspark {
#include "_ref/batch_gencfg_slice.inc"   // struct BatchGencfg { ... };
namespace cuda {
#include "_ref/logits_processor_slice.inc"  // the five kernels + template <typename T> void LogitsProcessor(...)
}  // namespace cuda
}  // namespace allspark

// score: device f32 [batch, vocab] (processed in place); in_ids: device int64 [batch, max_len]; the nine lists: device arrays [batch];
// ws: device scratch of max(batch * vocab * 4, batch * vocab * sizeof(float)) bytes (the reference uses it for the score copy, then the counts)
extern "C" int ref_logits_processor(float* score, const int64_t* in_ids, int batch, int max_len, int vocab, float* repetition, float* presence,
                                    float* frequency, int* ngram, int* min_length, int* eos, int* cur_len, int* input_len, int* suppress, void* ws,
                                    size_t ws_bytes, void* stream) {
  allspark::BatchGencfg g;
  g.batch_size = batch;
  g.repetition_penalty_list = repetition;
  g.presence_penalty_list = presence;
  g.frequency_penalty_list = frequency;
  g.no_repeat_ngram_size_list = ngram;
  g.min_length_list = min_length;
  g.eos_token_id_list = eos;
  g.cur_len_list = cur_len;
  g.input_len_list = input_len;
  g.suppress_repetition_in_generation_list = suppress;
  allspark::cuda::LogitsProcessor<float>(score, in_ids, batch, max_len, vocab, g, ws, ws_bytes, reinterpret_cast<hipStream_t>(stream));
  return (int)hipGetLastError();
}
