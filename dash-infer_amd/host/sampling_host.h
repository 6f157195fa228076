// sampling_host.h -- per-request sampling parameters of a batch on the device, for the operators that end a step
// (GenerateOp on HIP, DihipGreedy of the fused list).  GenerateOp::Reshape gathers top_k / top_p / temperature per request into
// device lists (generate_op.cpp:325-372); the seed feeds the per-request random stream (generate_op.cpp:486-490).
#pragma once
#include <cstring>
#include <limits>
#include <mutex>
#include <string>

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

// GenerateOp's logits processors and log-probability outputs for a batch (generate_op.cpp:239-312 build_batch_gencfg, :521-538 max_dec_ids
// + process_logits_launcher, :600-650 logprobs_launcher + UpdateProbs), in the two forms csrc/logits_proc.hip serves:
//   staged  (GenerateOp on HIP, operator by operator): every Forward copies the requests' host "generated_ids" into one [rows, max_len]
//           device tensor (what fill_max_dec_ids does device to device) and the log-probabilities come back to the host at once;
//   rows    (DihipGreedy under the model runner): the histories and the log-probability records are per-request device tensors the step
//           itself appends to (GenerateContext::history_dev / logprob_records_dev): a captured step replays with nothing from the host.
class LogitsProcParams {
 public:
  static constexpr int kStride = 10;                 // top_logprobs places per record (GetMaxTopLogprobs)
  static constexpr int kRecordWords = 1 + 2 * kStride;
  ~LogitsProcParams() {
    if (staged_) (void)hipEventSynchronize(staged_);  // (a staging copy may still read the pinned blocks)
    if (host_) (void)hipHostFree(host_);
    if (hist_host_) (void)hipHostFree(hist_host_);
    if (out_host_) (void)hipHostFree(out_host_);
    if (staged_) (void)hipEventDestroy(staged_);
  }
  bool any_processors() const { return any_proc_; }
  bool any_logprobs() const { return any_lp_; }
  int top_n() const { return top_n_; }
  // Reshape time.  rows_form: histories / records are the requests' own device tensors (their absence where a request asks is an error)
  AsStatus Gather(const RuntimeContext* rt, int rows, int vocab, int max_len, bool rows_form, hipStream_t s, std::string* why) {
    rows_ = rows;
    vocab_ = vocab;
    max_len_ = std::max(max_len, 1);
    any_proc_ = any_lp_ = false;
    top_n_ = 0;
    for (int i = 0; i < rows; ++i) {
      const GenerateContext* gc = rt->is_context ? rt->GetContextGenCtx() : rt->GetGenCtx(i);
      const GenerateConfig& g = gc->gen_cfg;
      any_proc_ = any_proc_ || g.has_logits_processors();
      if (g.logprobs) {
        if (g.top_logprobs < 0 || g.top_logprobs > kStride) return Say(why, "top_logprobs must be in [0, 10] (as_engine.cpp:2153)");
        any_lp_ = true;
        top_n_ = std::max(top_n_, g.top_logprobs);
      }
      if (!(g.repetition_penalty > 0.f)) return Say(why, "repetition_penalty must be positive");
    }
    if (!any_proc_ && !any_lp_) return AsStatus::ALLSPARK_SUCCESS;
    if (staged_ && hipEventSynchronize(staged_) != hipSuccess) return AsStatus::ALLSPARK_RUNTIME_ERROR;
    if ((size_t)rows > cap_) {
      if (host_) (void)hipHostFree(host_);
      cap_ = std::max<size_t>(rows, 32);
      if (hipHostMalloc((void**)&host_, Bytes(), hipHostMallocDefault) != hipSuccess) return AsStatus::ALLSPARK_MEMORY_ERROR;
      dev_ = std::make_unique<AsTensor>("logits_proc.params", DeviceType::HIP, INT8, Shape{(int64_t)Bytes()});
      if (!dev_->GetDataPtr()) return AsStatus::ALLSPARK_MEMORY_ERROR;
    }
    if (!staged_ && hipEventCreateWithFlags(&staged_, hipEventDisableTiming) != hipSuccess) return AsStatus::ALLSPARK_RUNTIME_ERROR;
    for (int i = 0; i < rows; ++i) {
      const GenerateContext* gc = rt->is_context ? rt->GetContextGenCtx() : rt->GetGenCtx(i);
      const GenerateConfig& g = gc->gen_cfg;
      H<void*>(0)[i] = nullptr;
      H<void*>(1)[i] = nullptr;
      if (rows_form) {
        if (g.has_logits_processors()) {
          if (!gc->history_dev || !gc->history_dev->GetDataPtr() || gc->history_dev->GetSizeInByte() < (size_t)max_len_ * sizeof(int64_t))
            return Say(why, "a request with logits processors has no device-resident token history (adopted requests / a cached prefix carry none)");
          H<void*>(0)[i] = gc->history_dev->GetDataPtr();
        }
        if (g.logprobs) {
          if (!gc->logprob_records_dev || !gc->logprob_records_dev->GetDataPtr()) return Say(why, "a request with logprobs has no record tensor");
          H<void*>(1)[i] = gc->logprob_records_dev->GetDataPtr();
        }
      }
      F(0)[i] = g.repetition_penalty;
      F(1)[i] = g.frequency_penalty;
      F(2)[i] = g.presence_penalty;
      I(0)[i] = g.no_repeat_ngram_size;
      I(1)[i] = g.min_length;
      I(2)[i] = g.eos_token_id;
      I(3)[i] = g.suppress_repetition_in_generation ? 1 : 0;
      I(4)[i] = gc->input_len;
      I(5)[i] = gc->step + gc->in_length_bias;  // cur_len, generate_op.cpp:277 (the rows form reads the device's length instead in the decoder phase)
    }
    if (any_proc_) {
      const int64_t need = (int64_t)dihip_logits_processor_workspace_bytes(rows, vocab);
      if (!count_ || (int64_t)count_->GetSizeInByte() < need) count_ = std::make_unique<AsTensor>("logits_proc.count", DeviceType::HIP, INT8, Shape{need});
      if (!count_->GetDataPtr()) return AsStatus::ALLSPARK_MEMORY_ERROR;
      if (!rows_form) {
        const int64_t hb = (int64_t)rows * max_len_ * (int64_t)sizeof(int64_t);
        if (!hist_ || (int64_t)hist_->GetSizeInByte() < hb) hist_ = std::make_unique<AsTensor>("logits_proc.max_dec_ids", DeviceType::HIP, INT8, Shape{hb});
        if (!hist_->GetDataPtr()) return AsStatus::ALLSPARK_MEMORY_ERROR;
        if (hist_host_cap_ < (size_t)hb) {
          if (hist_host_) (void)hipHostFree(hist_host_);
          hist_host_cap_ = 0;
          if (hipHostMalloc((void**)&hist_host_, (size_t)hb, hipHostMallocDefault) != hipSuccess) return AsStatus::ALLSPARK_MEMORY_ERROR;
          hist_host_cap_ = (size_t)hb;
        }
      }
    }
    if (any_lp_) {
      const int64_t wb = (int64_t)std::max<size_t>(dihip_logprobs_workspace_bytes(rows, vocab, top_n_), 8);
      if (!lp_ws_ || (int64_t)lp_ws_->GetSizeInByte() < wb) lp_ws_ = std::make_unique<AsTensor>("logits_proc.logprobs_ws", DeviceType::HIP, INT64, Shape{(wb + 7) / 8});
      if (!lp_ws_->GetDataPtr()) return AsStatus::ALLSPARK_MEMORY_ERROR;
    }
    if (any_lp_ && !rows_form) {
      const int64_t ob = (int64_t)rows * kRecordWords * 4;
      if (!out_ || (int64_t)out_->GetSizeInByte() < ob) {
        out_ = std::make_unique<AsTensor>("logits_proc.logprobs", DeviceType::HIP, INT8, Shape{ob});
        if (out_host_) (void)hipHostFree(out_host_);
        out_host_ = nullptr;
        if (!out_->GetDataPtr() || hipHostMalloc((void**)&out_host_, (size_t)ob, hipHostMallocDefault) != hipSuccess) return AsStatus::ALLSPARK_MEMORY_ERROR;
      }
    }
    return Upload(s);
  }
  // staged form, every Forward: the requests' host histories (request->interim["generated_ids"], [1, cur_len]) -> max_dec_ids, cur_len list
  AsStatus StageHistory(const RuntimeContext* rt, hipStream_t s, std::string* why) {
    if (hipEventSynchronize(staged_) != hipSuccess) return AsStatus::ALLSPARK_RUNTIME_ERROR;
    for (int i = 0; i < rows_; ++i) {
      const GenerateContext* gc = rt->is_context ? rt->GetContextGenCtx() : rt->GetGenCtx(i);
      int64_t* dst = hist_host_ + (size_t)i * max_len_;
      int cur = 0;
      if (gc->request) {
        auto it = gc->request->interim.find("generated_ids");
        if (it != gc->request->interim.end() && it->second->GetDataType() == INT64) {
          cur = (int)std::min<int64_t>({(int64_t)(gc->step + gc->in_length_bias), it->second->Count(), (int64_t)max_len_});
          std::memcpy(dst, it->second->GetDataPtr(), (size_t)std::max(cur, 0) * sizeof(int64_t));
        }
      }
      if (cur <= 0 && gc->gen_cfg.has_logits_processors() && gc->step + gc->in_length_bias > 0)
        return Say(why, "a request with logits processors carries no \"generated_ids\" (PreProcessId creates it, generate_impl_gpu.hpp:292-293)");
      I(5)[i] = std::max(cur, 0);
    }
    if (hipMemcpyAsync(hist_->GetDataPtr(), hist_host_, (size_t)rows_ * max_len_ * sizeof(int64_t), hipMemcpyHostToDevice, s) != hipSuccess)
      return AsStatus::ALLSPARK_RUNTIME_ERROR;
    return Upload(s);
  }
  // rows form, context phase: cur_len of the one request = the prompt's length (the history holds the prompt; nothing is appended)
  AsStatus StageCurLen(int cur_len, hipStream_t s) {
    if (hipEventSynchronize(staged_) != hipSuccess) return AsStatus::ALLSPARK_RUNTIME_ERROR;
    for (int i = 0; i < rows_; ++i) I(5)[i] = cur_len;
    return Upload(s);
  }
  AsStatus RunStaged(float* logits, hipStream_t s) {
    return FromDihip(dihip_logits_processor(s, logits, rows_, vocab_, (const int64_t*)hist_->GetDataPtr(), max_len_, D<int>(I(5)), D<int>(I(4)), D<float>(F(0)),
                                            D<float>(F(1)), D<float>(F(2)), D<int>(I(0)), D<int>(I(1)), D<int>(I(2)), D<int>(I(3)), count_->GetDataPtr(),
                                            count_->GetSizeInByte()));
  }
  // append_ids: the step's input ids (decoder phase) or null; cur_len_dev: the device-resident lengths after this step's token, or null: the staged list
  AsStatus RunRows(float* logits, const int64_t* append_ids, const uint32_t* cur_len_dev, hipStream_t s) {
    return FromDihip(dihip_logits_processor_rows(s, logits, rows_, vocab_, (int64_t* const*)D<void*>(H<void*>(0)), append_ids, max_len_,
                                                 cur_len_dev ? (const int*)cur_len_dev : D<int>(I(5)), D<int>(I(4)), D<float>(F(0)), D<float>(F(1)),
                                                 D<float>(F(2)), D<int>(I(0)), D<int>(I(1)), D<int>(I(2)), D<int>(I(3)), count_->GetDataPtr(),
                                                 count_->GetSizeInByte()));
  }
  AsStatus LogprobsRows(const float* logits, const int64_t* chosen, const uint32_t* position, int bias, hipStream_t s) {
    return FromDihip(dihip_logprobs_records(s, logits, rows_, vocab_, chosen, top_n_, kStride, (float* const*)D<void*>(H<void*>(1)), position, bias, max_len_,
                                            lp_ws_->GetDataPtr(), lp_ws_->GetSizeInByte()));
  }
  // staged form: compute, fetch, and append to the requests' lists as UpdateProbs does (generate_op.cpp:36-57; rank 0 only)
  AsStatus LogprobsStaged(const RuntimeContext* rt, const float* logits, const int64_t* chosen, bool publish, hipStream_t s) {
    float* o = (float*)out_->GetDataPtr();
    AS_CHECK_STATUS(FromDihip(dihip_logprobs(s, logits, rows_, vocab_, chosen, top_n_, kStride, o, o + rows_, (int*)(o + rows_ + (size_t)rows_ * kStride),
                                             lp_ws_->GetDataPtr(), lp_ws_->GetSizeInByte())));
    if (hipMemcpyAsync(out_host_, o, (size_t)rows_ * kRecordWords * 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
      return AsStatus::ALLSPARK_RUNTIME_ERROR;
    if (!publish) return AsStatus::ALLSPARK_SUCCESS;
    const float* tok = out_host_;
    const float* val = out_host_ + rows_;
    const int* idx = (const int*)(out_host_ + rows_ + (size_t)rows_ * kStride);
    for (int i = 0; i < rows_; ++i) {
      GenerateContext* gc = rt->is_context ? rt->GetContextGenCtx() : rt->GetGenCtx(i);
      if (!gc->gen_cfg.logprobs || !gc->request) continue;
      std::vector<std::pair<int, float>> top;
      for (int k = 0; k < gc->gen_cfg.top_logprobs; ++k) top.emplace_back(idx[(size_t)i * kStride + k], val[(size_t)i * kStride + k]);
      std::lock_guard<std::mutex> g(gc->request->queue_mu);
      gc->request->log_probs_list.push_back(std::move(top));
      gc->request->token_logprobs_list.push_back(tok[i]);
    }
    return AsStatus::ALLSPARK_SUCCESS;
  }

 private:
  // one pinned block / one device block: 2 pointer lists, 3 float lists, 6 int lists of cap_ entries each
  size_t Bytes() const { return cap_ * (2 * sizeof(void*) + 3 * sizeof(float) + 6 * sizeof(int)) + 64; }
  template <typename T>
  T* H(int k) const { return reinterpret_cast<T*>(host_) + (size_t)k * cap_; }
  float* F(int k) const { return reinterpret_cast<float*>(host_ + 2 * cap_ * sizeof(void*)) + (size_t)k * cap_; }
  int* I(int k) const { return reinterpret_cast<int*>(host_ + 2 * cap_ * sizeof(void*) + 3 * cap_ * sizeof(float)) + (size_t)k * cap_; }
  template <typename T>
  const T* D(const T* host_ptr) const {  // the device address of a host list
    return reinterpret_cast<const T*>((const char*)dev_->GetDataPtr() + ((const char*)host_ptr - host_));
  }
  AsStatus Upload(hipStream_t s) {
    if (hipMemcpyAsync(dev_->GetDataPtr(), host_, Bytes() - 64, hipMemcpyHostToDevice, s) != hipSuccess || hipEventRecord(staged_, s) != hipSuccess)
      return AsStatus::ALLSPARK_RUNTIME_ERROR;
    return AsStatus::ALLSPARK_SUCCESS;
  }
  static AsStatus Say(std::string* why, const char* msg) {
    if (why) *why = msg;
    return AsStatus::ALLSPARK_PARAM_ERROR;
  }
  char* host_ = nullptr;
  std::unique_ptr<AsTensor> dev_, count_, hist_, out_, lp_ws_;
  int64_t* hist_host_ = nullptr;
  size_t hist_host_cap_ = 0;
  float* out_host_ = nullptr;
  size_t cap_ = 0;
  int rows_ = 0, vocab_ = 0, max_len_ = 1, top_n_ = 0;
  bool any_proc_ = false, any_lp_ = false;
  hipEvent_t staged_ = nullptr;
};

}  // namespace allspark
