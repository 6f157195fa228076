// sampling_host.h -- per-request sampling parameters of a batch on the device, for the operators that end a step
// (GenerateOp on HIP, DihipGreedy of the fused list).  GenerateOp::Reshape gathers top_k / top_p / temperature per request into
// device lists (generate_op.cpp:325-372); the seed feeds the per-request random stream (generate_op.cpp:486-490).
#pragma once
#include <limits>

#include "dashinfer_hip.h"
#include "operator.h"

namespace allspark {

class SamplingParams {
 public:
  ~SamplingParams() {
    if (host_) (void)hipHostFree(host_);
    if (staged_) (void)hipEventDestroy(staged_);
  }
  bool any_sampling() const { return any_; }
  int wide_rows() const { return wide_; }  // rows with top_k == 0 or > 1024 (dihip_sample_rows launches the wide kernel only when there are any)
  // Reshape time: the requests of this forward (context: the one being prefilled)
  AsStatus Gather(const RuntimeContext* rt, int rows, hipStream_t s) {
    rows_ = rows;
    any_ = false;
    wide_ = 0;
    const size_t per = sizeof(int) + 2 * sizeof(float) + sizeof(unsigned long long) + sizeof(uint32_t);
    // the previous forward's staging copy reads host_: it must have completed before the buffer is freed or rewritten
    if (staged_ && hipEventSynchronize(staged_) != hipSuccess) return AsStatus::ALLSPARK_RUNTIME_ERROR;
    if ((size_t)rows > cap_) {
      if (host_) (void)hipHostFree(host_);
      cap_ = std::max<size_t>(rows, 32);
      if (hipHostMalloc((void**)&host_, cap_ * per + 64, hipHostMallocDefault) != hipSuccess) return AsStatus::ALLSPARK_MEMORY_ERROR;
      dev_ = std::make_unique<AsTensor>("sampling.params", DeviceType::HIP, INT8, Shape{(int64_t)(cap_ * per + 64)});
      if (!dev_->GetDataPtr()) return AsStatus::ALLSPARK_MEMORY_ERROR;
    }
    if (!staged_ && hipEventCreateWithFlags(&staged_, hipEventDisableTiming) != hipSuccess) return AsStatus::ALLSPARK_RUNTIME_ERROR;
    // layout: seeds [cap] (8-byte aligned first), top_k [cap], top_p [cap], temperature [cap], position [cap]
    auto* seed = reinterpret_cast<unsigned long long*>(host_);
    auto* tk = reinterpret_cast<int*>(seed + cap_);
    auto* tp = reinterpret_cast<float*>(tk + cap_);
    auto* tt = tp + cap_;
    for (int i = 0; i < rows; ++i) {
      const GenerateContext* gc = rt->is_context ? rt->GetContextGenCtx() : rt->GetGenCtx(i);
      const GenerateConfig& g = gc->gen_cfg;
      if (!(g.temperature >= std::numeric_limits<float>::min())) return AsStatus::ALLSPARK_PARAM_ERROR;  // generate_op.cpp:357-362
      // top_k == 0 is "the whole vocabulary" (real_k = vocab_size_, generate_op.cpp:338-339: pure top-p sampling) and any k is served, as
      // the reference's default build does: rows with k == 0 or k > 1024 take the sort-free wide kernel (csrc/sample.hip, round 6)
      if (g.top_k < 0) return AsStatus::ALLSPARK_PARAM_ERROR;
      wide_ += g.top_k == 0 || g.top_k > 1024;
      seed[i] = g.seed;
      tk[i] = g.top_k;
      tp[i] = g.top_p;
      tt[i] = g.temperature;
      any_ = any_ || g.top_k != 1;
    }
    if (hipMemcpyAsync(dev_->GetDataPtr(), host_, cap_ * per, hipMemcpyHostToDevice, s) != hipSuccess || hipEventRecord(staged_, s) != hipSuccess)
      return AsStatus::ALLSPARK_RUNTIME_ERROR;
    return AsStatus::ALLSPARK_SUCCESS;
  }
  // positions (index of the token being sampled, per row) staged from the host for this forward
  AsStatus StagePositions(const RuntimeContext* rt, int seq_len, hipStream_t s) {
    if (hipEventSynchronize(staged_) != hipSuccess) return AsStatus::ALLSPARK_RUNTIME_ERROR;
    uint32_t* pos = host_pos();
    for (int i = 0; i < rows_; ++i) {
      const GenerateContext* gc = rt->is_context ? rt->GetContextGenCtx() : rt->GetGenCtx(i);
      // tokens in the sequence once this forward's rows are cached: the cache's own length after Alloc (context: prefix + prompt,
      // decoder: step + 1 -- what "dihip.new_seq_lens" holds on the device), else step + rows of this forward
      pos[i] = gc->virtual_k_cache && gc->virtual_k_cache->GetLayerNum() > 0 ? (uint32_t)gc->virtual_k_cache->GetSeqLength(0)
                                                                               : (uint32_t)(gc->step + seq_len);
    }
    if (hipMemcpyAsync(dev_pos(), pos, (size_t)rows_ * sizeof(uint32_t), hipMemcpyHostToDevice, s) != hipSuccess ||
        hipEventRecord(staged_, s) != hipSuccess)
      return AsStatus::ALLSPARK_RUNTIME_ERROR;
    return AsStatus::ALLSPARK_SUCCESS;
  }
  const unsigned long long* seed() const { return reinterpret_cast<const unsigned long long*>(dev_->GetDataPtr()); }
  const int* top_k() const { return reinterpret_cast<const int*>(seed() + cap_); }
  const float* top_p() const { return reinterpret_cast<const float*>(top_k() + cap_); }
  const float* temperature() const { return top_p() + cap_; }
  uint32_t* dev_pos() const { return reinterpret_cast<uint32_t*>(const_cast<float*>(temperature() + cap_)); }

 private:
  uint32_t* host_pos() const {
    auto* seed = reinterpret_cast<unsigned long long*>(host_);
    return reinterpret_cast<uint32_t*>(reinterpret_cast<float*>(reinterpret_cast<int*>(seed + cap_) + cap_) + 2 * cap_);
  }
  char* host_ = nullptr;
  std::unique_ptr<AsTensor> dev_;
  size_t cap_ = 0;
  int rows_ = 0;
  bool any_ = false;
  int wide_ = 0;
  hipEvent_t staged_ = nullptr;
};

}  // namespace allspark
