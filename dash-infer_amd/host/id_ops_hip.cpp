// id_ops_hip.cpp -- the graph-head / graph-tail operators of the reference's generation graph on DeviceType::HIP:
//   TransMask      general/transmask/transmask_op.cpp: builds the [batch, seq, seq] attention mask for the context phase.  On
//                  the reference's GPU path with a flash / xformer prefill it is a no-op (Reshape and Forward return at once,
//                  transmask_op.cpp:101-107,131-137); this backend's prefill attention applies the causal mask in the kernel
//                  (csrc/prefill_attn.hip) -- same: attributes are validated, nothing is computed.
//   PreProcessId   generate_opt/preprocess_id/preprocess_id_op.cpp:32-80: at the start of a request, "generated_ids" (host,
//                  INT64 [1, max_length], shaped like input_ids) <- input_ids; device copies "generated_ids_gpu" and
//                  "new_input_ids_gpu" of the request's interim tensors.
//   UpdateId       generate_opt/update_id/update_id_op.cpp:40-156: after a step, per request: stop checks (length limit,
//                  eos with early_stopping, stop words) and the newly generated token into the request's queue (rank 0).
//   PostProcessId  generate_opt/postprocess_id/postprocess_id_op.cpp:27-31: no-op.
// Host state only (the Request / GenerateContext slices restated in as_types.h); the device copies are plain hipMemcpyAsync.
#include <cstring>

#include "dashinfer_hip.h"
#include "operator.h"

namespace allspark {

namespace {
hipStream_t stream_of(const DeviceContext* ctx) { return static_cast<const HIPContext*>(ctx)->GetStream(); }
const char* attr_ptr(const OperatorProto& p, const char* k) {
  auto it = p.attr.find(k);
  return it == p.attr.end() ? nullptr : it->second.c_str();
}
}  // namespace

class TransMaskHIP : public AsOperator {
 public:
  explicit TransMaskHIP(const std::string& t = "") : AsOperator(t) {}
  AsStatus Init(const OperatorProto& op_proto, const DeviceContext& ctx, const TensorMap& weights_map, TensorMap* tensor_map) override {
    AS_CHECK_STATUS(AsOperator::Init(op_proto, ctx, weights_map, tensor_map));
    bool seq_mask = false, blank = false;
    if (const char* p = attr_ptr(op_proto, "sequence_mask")) seq_mask = *(const bool*)p;
    if (const char* p = attr_ptr(op_proto, "blank")) blank = *(const bool*)p;
    if (seq_mask && blank) return AsStatus::ALLSPARK_RUNTIME_ERROR;  // transmask_op.cpp:64-68
    tensor_map_->at(out_names_[0])->SetDataType(FLOAT32);
    if (out_names_.size() > 1) tensor_map_->at(out_names_[1])->SetDataType(INT32);
    return AsStatus::ALLSPARK_SUCCESS;
  }
  AsStatus Reshape() override { return AsStatus::ALLSPARK_SUCCESS; }  // flash-style prefill: the mask is never materialised
  AsStatus Forward() override { return AsStatus::ALLSPARK_SUCCESS; }
};
REGISTER_OP(TransMask, HIP, TransMaskHIP)

class PreProcessIdHIP : public AsOperator {
 public:
  explicit PreProcessIdHIP(const std::string& t = "") : AsOperator(t) {}
  AsStatus Reshape(RuntimeContext*) override { return AsStatus::ALLSPARK_SUCCESS; }
  AsStatus Forward(RuntimeContext* rt) override {
    GenerateContext* gc = rt->GetContextGenCtx();
    if (!gc || !gc->request) return AsStatus::ALLSPARK_PARAM_ERROR;
    Request& rq = *gc->request;
    auto in = rq.inputs.find("input_ids");
    if (in == rq.inputs.end() || in->second->GetDataType() != INT64 || in->second->GetDeviceType() != CPU) return AsStatus::ALLSPARK_PARAM_ERROR;
    const AsTensor& ids = *in->second;
    const int64_t cap = std::max<int64_t>(ctx_->GetModelMaxLength(), ids.Count());
    // local generated_ids: capacity [1, max_length], shaped like input_ids, host copy
    auto gen = std::make_shared<AsTensor>("generated_ids", CPU, INT64, Shape{1, cap});
    AS_CHECK_STATUS(gen->SetShape(Shape(ids.GetShape())));
    std::memcpy(gen->GetDataPtr(), ids.GetDataPtr(), ids.GetSizeInByte());
    rq.interim["generated_ids"] = gen;
    hipStream_t s = stream_of(ctx_);
    auto to_device = [&](const std::string& name, int64_t capacity) -> AsStatus {
      const AsTensor& src = *rq.interim.at(name);
      auto dev = std::make_shared<AsTensor>(name + "_gpu", DeviceType::HIP, src.GetDataType(), Shape{1, std::max<int64_t>(capacity, src.Count())});
      if (!dev->GetDataPtr()) return AsStatus::ALLSPARK_MEMORY_ERROR;
      AS_CHECK_STATUS(dev->SetShape(Shape(src.GetShape())));
      if (hipMemcpyAsync(dev->GetDataPtr(), src.GetDataPtr(), src.GetSizeInByte(), hipMemcpyHostToDevice, s) != hipSuccess)
        return AsStatus::ALLSPARK_RUNTIME_ERROR;
      rq.interim[name + "_gpu"] = dev;
      return AsStatus::ALLSPARK_SUCCESS;
    };
    AS_CHECK_STATUS(to_device("generated_ids", cap));
    if (rq.interim.count("new_input_ids")) AS_CHECK_STATUS(to_device("new_input_ids", 0));  // (a prefix-cache hit leaves only the new tokens)
    if (hipStreamSynchronize(s) != hipSuccess) return AsStatus::ALLSPARK_RUNTIME_ERROR;     // the host tensors may be pageable
    return AsStatus::ALLSPARK_SUCCESS;
  }
};
REGISTER_OP(PreProcessId, HIP, PreProcessIdHIP)

class UpdateIdHIP : public AsOperator {
 public:
  explicit UpdateIdHIP(const std::string& t = "") : AsOperator(t) {}
  AsStatus Reshape(RuntimeContext*) override { return AsStatus::ALLSPARK_SUCCESS; }
  AsStatus Forward(RuntimeContext* rt) override {
    if (GetOpName() == "update_id_first") return AsStatus::ALLSPARK_SUCCESS;  // update_id_op.cpp:144
    if (rt->is_context) return One(rt->GetContextGenCtx(), true);
    for (int i = 0; i < rt->GetGenCtxListSize(); ++i) AS_CHECK_STATUS(One(rt->GetGenCtx(i), false));
    return AsStatus::ALLSPARK_SUCCESS;
  }

 private:
  static bool StopWords(int generated_len, const int64_t* ids, bool* gen_over, const std::vector<std::vector<int64_t>>& words) {
    bool matched = false;  // update_id_op.cpp:16-40 (batch 1)
    for (const auto& w : words)
      if (generated_len > (int)w.size() && std::memcmp(ids + generated_len - w.size(), w.data(), w.size() * sizeof(int64_t)) == 0) {
        matched = true;
        break;
      }
    gen_over[0] |= matched;
    return gen_over[0];
  }
  static void CheckFinish(GenerateContext* gc, const int64_t* ids) {  // update_id_op.cpp:42-75
    if (gc->finish) return;
    if (gc->gen_cfg.max_length > 0 && gc->step + gc->in_length_bias >= gc->gen_cfg.max_length - 1) {
      gc->finish = true;
      return;
    }
    if (gc->gen_cfg.early_stopping && ids[gc->step + gc->in_length_bias] == (int64_t)gc->gen_cfg.eos_token_id) {
      gc->finish = true;
      return;
    }
    if (gc->generate_method == 0 && !gc->gen_cfg.stop_words_ids.empty() &&
        StopWords(gc->step + gc->in_length_bias + 1, ids, gc->gen_over, gc->gen_cfg.stop_words_ids))
      gc->finish = true;
  }
  AsStatus One(GenerateContext* gc, bool is_context) {
    if (!gc || !gc->request) return AsStatus::ALLSPARK_PARAM_ERROR;
    auto it = gc->request->interim.find("generated_ids");
    if (it == gc->request->interim.end()) return AsStatus::ALLSPARK_PARAM_ERROR;
    const AsTensor& t = *it->second;
    const int64_t* ids = static_cast<const int64_t*>(t.GetDataPtr());
    const int pos = gc->step + (is_context ? gc->in_length_bias : 0);
    if (pos < 0 || gc->step + gc->in_length_bias >= t.Count()) return AsStatus::ALLSPARK_EXCEED_LIMIT_ERROR;
    CheckFinish(gc, ids);
    if (ctx_->GetRank() == 0) gc->request->enqueue(ids[pos]);  // copy_generated_ids, update_id_op.cpp:77-92
    return AsStatus::ALLSPARK_SUCCESS;
  }
};
REGISTER_OP(UpdateId, HIP, UpdateIdHIP)

class PostProcessIdHIP : public AsOperator {
 public:
  explicit PostProcessIdHIP(const std::string& t = "") : AsOperator(t) {}
  AsStatus Reshape(RuntimeContext*) override { return AsStatus::ALLSPARK_SUCCESS; }
  AsStatus Forward(RuntimeContext*) override { return AsStatus::ALLSPARK_SUCCESS; }
};
REGISTER_OP(PostProcessId, HIP, PostProcessIdHIP)

}  // namespace allspark
