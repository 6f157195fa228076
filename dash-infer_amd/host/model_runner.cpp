// model_runner.cpp -- see model_runner.h.
#include "model_runner.h"

#include <string>

#include "dashinfer_hip.h"

#include <algorithm>

namespace allspark {

namespace {
const char* kOldLens = "dihip.old_seq_lens";
const char* kNewLens = "dihip.new_seq_lens";
}  // namespace

HipModelRunner::HipModelRunner(HIPContext* ctx, TensorMap* tensors, TensorMap* weights, TensorMap* weights_buffer)
    : ctx_(ctx), tensors_(tensors), weights_(weights), weights_buffer_(weights_buffer) {}

HipModelRunner::~HipModelRunner() {
  DropGraph();
  ops_.clear();
  if (ids_pinned_) (void)hipHostFree(ids_pinned_);
  if (lens_pinned_) (void)hipHostFree(lens_pinned_);
  if (block_err_pinned_) (void)hipHostFree(block_err_pinned_);
  if (prompt_pinned_) (void)hipHostFree(prompt_pinned_);
  if (staged_) (void)hipEventDestroy(staged_);
}

AsStatus HipModelRunner::Fail(AsStatus st, const std::string& what) {
  err_ = what;
  return st;
}

void HipModelRunner::DropGraph() {
  if (graph_exec_) (void)hipGraphExecDestroy(graph_exec_);
  if (graph_) (void)hipGraphDestroy(graph_);
  graph_exec_ = nullptr;
  graph_ = nullptr;
}

AsStatus HipModelRunner::Build(const std::vector<OperatorProto>& graph, bool fuse) {
  if (graph.empty() || graph.front().inputs.empty() || graph.back().outputs.empty()) return Fail(AsStatus::ALLSPARK_PARAM_ERROR, "empty graph");
  protos_ = fuse ? FuseDecoderGraph(graph, *ctx_, &fusion_) : graph;
  if (!fuse) {
    fusion_ = FusionReport();
    fusion_.ops_before = fusion_.ops_after = (int)graph.size();
    fusion_.why = "fusion not requested";
  }
  fused_ = fusion_.fused && fusion_.device_resident;  // device-resident step state (and graph replay) needs the fused tail
  ids_in_name_ = protos_.front().inputs[0];
  ctx_->SetInputIdsName(ids_in_name_);
  // The converter's gen_graph writes the sampled ids back into the tensor the embedding reads ("dec_ids", qwen_v15.py:440-447:
  // gen_op.outputs[0].CopyFrom(preprocess_ids.outputs[0])).  Here the input name is re-bound per phase (prompt rows / the running
  // batch's ids) while the sampling operator's output is one fixed device buffer: the output gets a name of its own.
  if (protos_.back().outputs[0] == ids_in_name_) protos_.back().outputs[0] += ".next";
  ids_out_name_ = protos_.back().outputs[0];
  const int64_t mb = std::max(1, ctx_->GetModelMaxBatch());
  // the ids the sampling operator writes are the next step's input: one device buffer, sized for the largest batch up front so that
  // its address never changes (the captured step reads and writes it)
  (*tensors_)[ids_out_name_] = std::make_shared<AsTensor>(ids_out_name_, DeviceType::HIP, INT64, Shape{mb, 1});
  (*tensors_)[kOldLens] = std::make_shared<AsTensor>(kOldLens, DeviceType::HIP, INT32, Shape{mb});
  (*tensors_)[kNewLens] = std::make_shared<AsTensor>(kNewLens, DeviceType::HIP, INT32, Shape{mb});
  const int64_t ml = std::max(1, ctx_->GetModelMaxLength());
  prompt_dev_ = std::make_unique<AsTensor>("runner.prompt_ids", DeviceType::HIP, INT64, Shape{ml});
  if (!(*tensors_)[ids_out_name_]->GetDataPtr() || !(*tensors_)[kOldLens]->GetDataPtr() || !(*tensors_)[kNewLens]->GetDataPtr() ||
      !prompt_dev_->GetDataPtr())
    return Fail(AsStatus::ALLSPARK_MEMORY_ERROR, "runner state tensors");
  if (hipHostMalloc((void**)&ids_pinned_, 2 * mb * sizeof(int64_t), hipHostMallocDefault) != hipSuccess ||
      hipHostMalloc((void**)&lens_pinned_, 2 * mb * sizeof(int32_t), hipHostMallocDefault) != hipSuccess ||
      hipHostMalloc((void**)&block_err_pinned_, 64, hipHostMallocDefault) != hipSuccess ||
      hipEventCreateWithFlags(&staged_, hipEventDisableTiming) != hipSuccess)
    return Fail(AsStatus::ALLSPARK_MEMORY_ERROR, "pinned staging");
  BindIds(prompt_dev_->GetDataPtr(), 1, 1);
  // lengths on the device only with the fused list: its attention / sampling operators read and advance them there
  ctx_->SetLensOnDevice(fused_);
  RuntimeContext init_rt;
  for (const OperatorProto& p : protos_) {
    std::unique_ptr<AsOperator> op;
    try {
      op = OpFactory::getInstance().GetOperator({p.op_type, DeviceType::HIP})();  // model.cpp:265-287
    } catch (const AsException& e) {
      return Fail(AsStatus::ALLSPARK_PARAM_ERROR, std::string(e.what()) + " (" + p.op_type + ")");
    }
    const AsStatus st = op->CallInit(p, *ctx_, *weights_, *weights_buffer_, tensors_, &init_rt);
    if (st != AsStatus::ALLSPARK_SUCCESS) return Fail(st, "CallInit failed: " + p.op_type + " " + p.op_name);
    ops_.push_back(std::move(op));
  }
  dirty_ = true;
  return AsStatus::ALLSPARK_SUCCESS;
}

void HipModelRunner::BindIds(void* data, int64_t rows, int64_t cols) {
  (*tensors_)[ids_in_name_] = std::make_shared<AsTensor>(ids_in_name_, DeviceType::HIP, INT64, Shape{rows, cols}, data);
}

AsStatus HipModelRunner::ForEach(RuntimeContext* rt, int phase) {
  static const char* names[] = {"CallReshape", "CallAlloc", "CallForward"};
  for (size_t i = 0; i < ops_.size(); ++i) {
    const AsStatus st = phase == 0 ? ops_[i]->CallReshape(rt) : phase == 1 ? ops_[i]->CallAlloc(rt) : ops_[i]->CallForward(rt);
    if (st != AsStatus::ALLSPARK_SUCCESS)
      return Fail(st, std::string(names[phase]) + " failed: " + protos_[i].op_type + " " + protos_[i].op_name);
  }
  return AsStatus::ALLSPARK_SUCCESS;
}

AsStatus HipModelRunner::StartRequest(std::shared_ptr<GenerateContext> gc, const int64_t* prompt_host, int len, int64_t* first_id) {
  if (!gc || !prompt_host || len <= 0 || len > ctx_->GetModelMaxLength()) return Fail(AsStatus::ALLSPARK_PARAM_ERROR, "StartRequest: bad prompt");
  if ((int)running_.size() >= std::max(1, ctx_->GetModelMaxBatch())) return Fail(AsStatus::ALLSPARK_EXCEED_LIMIT_ERROR, "StartRequest: batch full");
  hipStream_t s = ctx_->GetStream();
  // the running batch's ids live in the buffer the context phase is about to overwrite row 0 of: fetch them first
  if (!running_.empty()) AS_CHECK_STATUS(Sync(nullptr));
  if ((size_t)len > prompt_cap_) {
    if (prompt_pinned_) (void)hipHostFree(prompt_pinned_);
    prompt_cap_ = std::max<size_t>(len, 1024);
    if (hipHostMalloc((void**)&prompt_pinned_, prompt_cap_ * sizeof(int64_t), hipHostMallocDefault) != hipSuccess)
      return Fail(AsStatus::ALLSPARK_MEMORY_ERROR, "prompt staging");
  }
  if (hipEventSynchronize(staged_) != hipSuccess) return Fail(AsStatus::ALLSPARK_RUNTIME_ERROR, "staging event");
  std::copy(prompt_host, prompt_host + len, prompt_pinned_);
  if (hipMemcpyAsync(prompt_dev_->GetDataPtr(), prompt_pinned_, (size_t)len * sizeof(int64_t), hipMemcpyHostToDevice, s) != hipSuccess ||
      hipEventRecord(staged_, s) != hipSuccess)
    return Fail(AsStatus::ALLSPARK_RUNTIME_ERROR, "prompt upload");
  gc->step = gc->prefix_len;  // model.cpp:532
  gc->input_len = len;
  // requests with logits processors / logprobs carry their token history and log-probability records on the device (the step appends
  // to both itself: csrc/logits_proc.hip, rows / records forms).  The history starts as the prompt.
  if (gc->gen_cfg.has_logits_processors()) {
    if (gc->prefix_len != 0) return Fail(AsStatus::ALLSPARK_PARAM_ERROR, "StartRequest: logits processors over a cached prefix (the prefix's ids are not part of this prompt)");
    gc->history_dev = std::make_shared<AsTensor>("history", DeviceType::HIP, INT64, Shape{(int64_t)std::max(ctx_->GetModelMaxLength(), len + 1)});
    if (!gc->history_dev->GetDataPtr() ||
        hipMemcpyAsync(gc->history_dev->GetDataPtr(), prompt_dev_->GetDataPtr(), (size_t)len * sizeof(int64_t), hipMemcpyDeviceToDevice, s) != hipSuccess)
      return Fail(AsStatus::ALLSPARK_MEMORY_ERROR, "StartRequest: token history");
  }
  if (gc->gen_cfg.logprobs) {
    const int64_t words = (int64_t)std::max(ctx_->GetModelMaxLength(), len + 1) * (1 + 2 * ctx_->GetMaxTopLogprobs());
    gc->logprob_records_dev = std::make_shared<AsTensor>("logprob_records", DeviceType::HIP, FLOAT32, Shape{words});
    if (!gc->logprob_records_dev->GetDataPtr()) return Fail(AsStatus::ALLSPARK_MEMORY_ERROR, "StartRequest: log-probability records");
  }
  RuntimeContext rt;
  rt.is_context = true;
  rt.current_batch = 0;
  rt.gen_ctx_list = {gc};
  BindIds(prompt_dev_->GetDataPtr(), 1, len);
  DropGraph();
  dirty_ = true;  // the operators are reshaped for the prompt now
  AS_CHECK_STATUS(ForEach(&rt, 0));
  AS_CHECK_STATUS(ForEach(&rt, 1));
  AS_CHECK_STATUS(ForEach(&rt, 2));
  int64_t id = 0;
  if (hipMemcpyAsync(ids_pinned_, (*tensors_)[ids_out_name_]->GetDataPtr(), sizeof(int64_t), hipMemcpyDeviceToHost, s) != hipSuccess ||
      hipStreamSynchronize(s) != hipSuccess)
    return Fail(AsStatus::ALLSPARK_RUNTIME_ERROR, "first id readback");
  id = ids_pinned_[0];
  gc->step += len;  // tokens in the cache after the context phase
  running_.push_back(std::move(gc));
  next_ids_.push_back(id);
  if (first_id) *first_id = id;
  return AsStatus::ALLSPARK_SUCCESS;
}

AsStatus HipModelRunner::AdoptRequest(std::shared_ptr<GenerateContext> gc, int64_t next_id) {
  if (!gc) return Fail(AsStatus::ALLSPARK_PARAM_ERROR, "AdoptRequest: null");
  if ((int)running_.size() >= std::max(1, ctx_->GetModelMaxBatch())) return Fail(AsStatus::ALLSPARK_EXCEED_LIMIT_ERROR, "AdoptRequest: batch full");
  if (!running_.empty() && !dirty_) AS_CHECK_STATUS(Sync(nullptr));
  running_.push_back(std::move(gc));
  next_ids_.push_back(next_id);
  dirty_ = true;
  return AsStatus::ALLSPARK_SUCCESS;
}

AsStatus HipModelRunner::StopRequest(int index) {
  if (index < 0 || index >= (int)running_.size()) return Fail(AsStatus::ALLSPARK_PARAM_ERROR, "StopRequest: no such request");
  if (!dirty_) AS_CHECK_STATUS(Sync(nullptr));  // the survivors' next ids are on the device
  running_.erase(running_.begin() + index);
  next_ids_.erase(next_ids_.begin() + index);
  dirty_ = true;
  return AsStatus::ALLSPARK_SUCCESS;
}

// Batch membership changed (or the operators were reshaped for a prompt): next ids and lengths to the device, operators reshaped
// for [batch, 1], the captured step dropped (model.cpp:1226-1246: Reshape only on such a change).
AsStatus HipModelRunner::PrepareBatch() {
  const int B = (int)running_.size();
  if (B == 0) return Fail(AsStatus::ALLSPARK_INVALID_CALL_ERROR, "no running request");
  hipStream_t s = ctx_->GetStream();
  if (hipEventSynchronize(staged_) != hipSuccess) return Fail(AsStatus::ALLSPARK_RUNTIME_ERROR, "staging event");
  for (int b = 0; b < B; ++b) {
    ids_pinned_[b] = next_ids_[b];
    lens_pinned_[b] = running_[b]->step;
    lens_pinned_[B + b] = running_[b]->step + 1;
  }
  void* ids_dev = (*tensors_)[ids_out_name_]->GetDataPtr();
  if (hipMemcpyAsync(ids_dev, ids_pinned_, (size_t)B * sizeof(int64_t), hipMemcpyHostToDevice, s) != hipSuccess ||
      hipMemcpyAsync((*tensors_)[kOldLens]->GetDataPtr(), lens_pinned_, (size_t)B * sizeof(int32_t), hipMemcpyHostToDevice, s) != hipSuccess ||
      hipMemcpyAsync((*tensors_)[kNewLens]->GetDataPtr(), lens_pinned_ + B, (size_t)B * sizeof(int32_t), hipMemcpyHostToDevice, s) != hipSuccess ||
      hipEventRecord(staged_, s) != hipSuccess)
    return Fail(AsStatus::ALLSPARK_RUNTIME_ERROR, "batch state upload");
  BindIds(ids_dev, B, 1);
  decode_rt_.is_context = false;
  decode_rt_.current_batch = 0;
  decode_rt_.gen_ctx_list = running_;
  DropGraph();
  AS_CHECK_STATUS(ForEach(&decode_rt_, 0));
  // the sampling operator's Reshape may have re-shaped the id tensor (same storage): re-bind the input view
  BindIds((*tensors_)[ids_out_name_]->GetDataPtr(), B, 1);
  if ((*tensors_)[ids_out_name_]->GetDataPtr() != ids_dev) return Fail(AsStatus::ALLSPARK_RUNTIME_ERROR, "the id tensor moved");
  dirty_ = false;
  return AsStatus::ALLSPARK_SUCCESS;
}

AsStatus HipModelRunner::DecodeSteps(int n, bool use_graph) {
  if (use_graph && !fused_) return Fail(AsStatus::ALLSPARK_INVALID_CALL_ERROR, "graph replay needs the fused operator list (" + fusion_.why + ")");
  hipStream_t s = ctx_->GetStream();
  // launch plans follow the running requests' length in buckets (HIPContext::PlanLength), not the engine's maximum length; a new bucket is a
  // batch change as far as the operators are concerned: ids back to the host, Reshape, a new captured step (once per `bucket` tokens)
  static const int bucket = [] {
    const char* e = getenv("DIHIP_PLAN_BUCKET");  // tokens; 0: plans for the maximum length (A/B)
    return e ? std::max(0, atoi(e)) : 512;
  }();
  for (int it = 0; it < n; ++it) {
    if (bucket > 0 && !running_.empty()) {
      int need = 1;
      for (const auto& gc : running_) need = std::max(need, gc->step + 1);
      const int want = std::min(ctx_->GetModelMaxLength(), (need + bucket - 1) / bucket * bucket);
      if (want != ctx_->PlanLength()) {
        if (!dirty_) AS_CHECK_STATUS(Sync(nullptr));  // (the next ids are on the device)
        ctx_->SetPlanLength(want);
        dirty_ = true;
      }
    }
    if (dirty_) AS_CHECK_STATUS(PrepareBatch());
    for (const auto& gc : running_)
      if (gc->step + 1 > ctx_->GetModelMaxLength()) return Fail(AsStatus::ALLSPARK_EXCEED_LIMIT_ERROR, "sequence length limit");
    AS_CHECK_STATUS(ForEach(&decode_rt_, 1));  // cache growth: host work only, plus a table upload when a span was claimed
    if (use_graph) {
      if (!graph_exec_) {
        if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) return Fail(AsStatus::ALLSPARK_RUNTIME_ERROR, "begin capture");
        const AsStatus st = ForEach(&decode_rt_, 2);
        const hipError_t e = hipStreamEndCapture(s, &graph_);
        if (st != AsStatus::ALLSPARK_SUCCESS) {
          DropGraph();
          return st;
        }
        if (e != hipSuccess || hipGraphInstantiate(&graph_exec_, graph_, nullptr, nullptr, 0) != hipSuccess) {
          DropGraph();
          return Fail(AsStatus::ALLSPARK_RUNTIME_ERROR, std::string("graph capture / instantiate: ") + hipGetErrorString(hipGetLastError()));
        }
      }
      if (hipGraphLaunch(graph_exec_, s) != hipSuccess) return Fail(AsStatus::ALLSPARK_RUNTIME_ERROR, "hipGraphLaunch");
    } else {
      AS_CHECK_STATUS(ForEach(&decode_rt_, 2));
    }
    for (auto& gc : running_) gc->step += 1;  // model.cpp:1320
    ++steps_since_sync_;
  }
  return AsStatus::ALLSPARK_SUCCESS;
}

AsStatus HipModelRunner::Rewind(int cached_len) {
  AS_CHECK_STATUS(Sync(nullptr));
  for (auto& gc : running_) gc->step = cached_len;
  if (dirty_) return AsStatus::ALLSPARK_SUCCESS;  // PrepareBatch uploads everything anyway
  hipStream_t s = ctx_->GetStream();
  const int B = (int)running_.size();
  for (int b = 0; b < B; ++b) {
    lens_pinned_[b] = cached_len;
    lens_pinned_[B + b] = cached_len + 1;
  }
  if (hipMemcpyAsync((*tensors_)[kOldLens]->GetDataPtr(), lens_pinned_, (size_t)B * sizeof(int32_t), hipMemcpyHostToDevice, s) != hipSuccess ||
      hipMemcpyAsync((*tensors_)[kNewLens]->GetDataPtr(), lens_pinned_ + B, (size_t)B * sizeof(int32_t), hipMemcpyHostToDevice, s) != hipSuccess ||
      hipStreamSynchronize(s) != hipSuccess)
    return Fail(AsStatus::ALLSPARK_RUNTIME_ERROR, "rewind: length upload");
  return AsStatus::ALLSPARK_SUCCESS;
}

AsStatus HipModelRunner::Sync(std::vector<int64_t>* ids) {
  hipStream_t s = ctx_->GetStream();
  const int B = (int)running_.size();
  const int64_t mb = std::max(1, ctx_->GetModelMaxBatch());
  if (B > 0 && !dirty_) {
    if (hipMemcpyAsync(ids_pinned_ + mb, (*tensors_)[ids_out_name_]->GetDataPtr(), (size_t)B * sizeof(int64_t), hipMemcpyDeviceToHost, s) != hipSuccess)
      return Fail(AsStatus::ALLSPARK_RUNTIME_ERROR, "id readback");
  }
  // the fused attention block's error word rides on the same synchronisation (a hand-off that timed out must not turn into
  // silently wrong tokens): the copy is enqueued beside the id readback, read after the one stream synchronise
  unsigned* blk_err = block_err_pinned_;
  auto blk = tensors_->find("dihip.attn_block_sync");
  const bool blk_live = blk != tensors_->end() && blk->second->GetDataPtr() && !ctx_->AttnBlockDisabled() && blk_err;
  if (blk_live) {
    *blk_err = 0u;
    if (dihip_decode_attn_block_status_async(s, blk->second->GetDataPtr(), blk_err) != 0) return Fail(AsStatus::ALLSPARK_RUNTIME_ERROR, "block status readback");
  }
  if (hipStreamSynchronize(s) != hipSuccess) return Fail(AsStatus::ALLSPARK_RUNTIME_ERROR, "stream synchronise");
  if (blk_live && *blk_err != 0u) {
    const unsigned code = *blk_err;
    // the step(s) since the last synchronise are invalid; the ids on the device are garbage: keep the host's last good ones, switch the
    // layers back to their three launches (Reshape re-decides, the captured step is dropped) and restore the buffer's invariants
    for (auto& gc : running_) gc->step -= std::min(gc->step, steps_since_sync_);  // back to the state of the last good synchronise
    steps_since_sync_ = 0;
    handoff_failed_ = true;
    ctx_->DisableAttnBlock();
    (void)dihip_decode_attn_block_reset(s, blk->second->GetDataPtr(), blk->second->GetSizeInByte());
    (void)hipStreamSynchronize(s);
    dirty_ = true;
    DropGraph();
    return Fail(AsStatus::ALLSPARK_RUNTIME_ERROR,
                "attention block: a bounded hand-off wait gave up (code " + std::to_string(code) +
                    "): the decode steps since the last synchronise are invalid; the launch chain serves from here "
                    "(were all workgroups resident? CU mask / another stream's kernel on this GPU)");
  }
  steps_since_sync_ = 0;
  handoff_failed_ = false;
  if (B > 0 && !dirty_)
    for (int b = 0; b < B; ++b) next_ids_[b] = ids_pinned_[mb + b];
  if (ids) *ids = next_ids_;
  return AsStatus::ALLSPARK_SUCCESS;
}

}  // namespace allspark
