// glue_ops_hip.cpp -- the remaining operators of the Qwen2 layer graph (python/pyhie/allspark/model/qwen_v15.py:187-388,
// model_base.py:690-703) on DeviceType::HIP, so that AsModel's operator look-up (csrc/core/model/model.cpp:265-287 ->
// OpFactory::GetOperator, operator.cpp:379-386) resolves every op type of the unmodified graph:
//   LayerNormNoBeta  csrc/core/operator/general/layernorm_nobeta/layernorm_nobeta_op.cpp   (weights [gamma], attr eps)
//   Rotary           general/rotary/rotary_op.cpp  (attrs num_heads, multi_query_group_num, rotary_base; positions from the
//                    RuntimeContext: gen_ctx->step per request in the decoder phase, step + t in the context phase)
//   Binary           general/binary/binary_op.cpp  (attr binary_type: ADD = 1, MUL = 2)
//   Unary / UnaryGLU general/unary/unary_op.cpp, general/unary_glu/unary_glu_op.cpp  (attr unary_type)
//   EmbeddingT5      general/embeddingT5/embeddingT5_op.cpp  (weights [word_embeddings]; ids INT64 [batch, seq])
//   GetLastLine      general/get_last_line/get_last_line.cpp
//   GenerateOp       generate_opt/generate/generate_op.cpp -- greedy (top_k = 1: arg-max) and the sampling half (per-request
//                    top_k / top_p / temperature / seed from gen_cfg, generate_op.cpp:325-372,472-600 -> dihip_sample); the
//                    logits processors (repetition / presence / frequency penalties, no-repeat n-grams, min length), logprobs
//                    and the json formatter stay the engine's
// Every operator binds tensors by name, infers its output type / shape like the reference op and only enqueues C-ABI calls
// (include/dashinfer_hip.h section 5) on the context's stream.  The decode step of the product (decoder.py, bench.py)
// never runs these as separate launches -- they ride in GEMV prologues / epilogues there; this file is the drop-in
// surface for the reference's own graph.
#include <cstdio>
#include <algorithm>
#include <cmath>

#include "dashinfer_hip.h"
#include "operator.h"
#include "sampling_host.h"

namespace allspark {

namespace {
hipStream_t stream_of(const DeviceContext* ctx) { return static_cast<const HIPContext*>(ctx)->GetStream(); }
const char* attr_ptr(const OperatorProto& p, const char* k) {
  auto it = p.attr.find(k);
  return it == p.attr.end() ? nullptr : it->second.c_str();
}
bool is_ft(DataType t) { return t == FLOAT16 || t == BFLOAT16 || t == FLOAT32; }
}  // namespace

// ------------------------------------------------------------------------------------------------ LayerNormNoBeta
class LayerNormNoBetaHIP : public AsOperator {
 public:
  explicit LayerNormNoBetaHIP(const std::string& t = "") : AsOperator(t) {}
  AsStatus Init(const OperatorProto& op_proto, const DeviceContext& ctx, const TensorMap& weights_map, TensorMap* tensor_map) override {
    AS_CHECK_STATUS(AsOperator::Init(op_proto, ctx, weights_map, tensor_map));
    if (weights_.size() != 1) return AsStatus::ALLSPARK_PARAM_ERROR;  // [gamma], layernorm_nobeta_op.cpp:60-63
    if (in_names_.size() != 1) return AsStatus::ALLSPARK_PARAM_ERROR;  // the bias input of the reference op is not used by this graph
    hidden_ = (int)weights_[0]->GetShape()[0];
    dtype_ = weights_[0]->GetDataType();
    if (!is_ft(dtype_)) return AsStatus::ALLSPARK_PARAM_ERROR;
    tensor_map_->at(out_names_[0])->SetDataType(dtype_);
    const char* e = attr_ptr(op_proto, "eps");
    if (!e) return AsStatus::ALLSPARK_PARAM_ERROR;  // :69-73
    eps_ = *(const float*)e;
    return AsStatus::ALLSPARK_SUCCESS;
  }
  AsStatus Reshape() override {
    Shape s = tensor_map_->at(in_names_[0])->GetShape();
    if (s.empty() || (int)s.back() != hidden_) return AsStatus::ALLSPARK_PARAM_ERROR;
    return tensor_map_->at(out_names_[0])->SetShape(std::move(s));
  }
  AsStatus Forward() override {
    AsTensor* x = tensor_map_->at(in_names_[0]).get();
    AsTensor* y = tensor_map_->at(out_names_[0]).get();
    if (x->GetDataType() != dtype_) return AsStatus::ALLSPARK_PARAM_ERROR;
    return FromDihip(dihip_rmsnorm(stream_of(ctx_), y->GetDataPtr(), x->GetDataPtr(), weights_[0]->GetDataPtr(), eps_,
                                   (int)(x->Count() / hidden_), hidden_, DihipDtype(dtype_)));
  }

 private:
  int hidden_ = 0;
  float eps_ = 1e-6f;
  DataType dtype_ = BFLOAT16;
};
REGISTER_OP(LayerNormNoBeta, HIP, LayerNormNoBetaHIP)

// ------------------------------------------------------------------------------------------------ Rotary
class RotaryHIP : public AsOperator {
 public:
  explicit RotaryHIP(const std::string& t = "") : AsOperator(t) {}
  ~RotaryHIP() override {
    if (pos_host_) (void)hipHostFree(pos_host_);
    if (staged_) (void)hipEventDestroy(staged_);
  }
  AsStatus Init(const OperatorProto& op_proto, const DeviceContext& ctx, const TensorMap& weights_map, TensorMap* tensor_map) override {
    AS_CHECK_STATUS(AsOperator::Init(op_proto, ctx, weights_map, tensor_map));
    // type inference at Init, as every reference operator does it (rotary_op.cpp:90-91): the NEXT operator's Init reads it
    tensor_map_->at(out_names_[0])->SetDataType(tensor_map_->at(in_names_[0])->GetDataType());
    const char* p = attr_ptr(op_proto, "num_heads");
    if (!p) return AsStatus::ALLSPARK_PARAM_ERROR;  // rotary_op.cpp:94-98
    num_heads_ = *(const int*)p;
    group_num_ = num_heads_;
    if ((p = attr_ptr(op_proto, "multi_query_group_num"))) group_num_ = *(const int*)p;
    if ((p = attr_ptr(op_proto, "rotary_base"))) base_ = *(const float*)p;
    // the variants this backend does not implement are refused at Init, never silently ignored
    if ((p = attr_ptr(op_proto, "rotary_type")) && *(const int*)p != 0) return AsStatus::ALLSPARK_PARAM_ERROR;
    if ((p = attr_ptr(op_proto, "rotary_pct")) && *(const float*)p != 1.0f) return AsStatus::ALLSPARK_PARAM_ERROR;
    if ((p = attr_ptr(op_proto, "invfreq_type")) && *(const int*)p != 0) return AsStatus::ALLSPARK_PARAM_ERROR;
    for (const char* k : {"ntk_model_embed", "logn_model_embedding", "mrope_section_size", "seqlen_extrapolation", "rope_ratio",
                          "original_max_position_embeddings", "use_weight"})
      if (attr_ptr(op_proto, k)) return AsStatus::ALLSPARK_PARAM_ERROR;
    size_per_head_ = ctx.GetSizePerHead();
    if (size_per_head_ <= 0 || size_per_head_ % 2 || num_heads_ <= 0 || group_num_ <= 0) return AsStatus::ALLSPARK_PARAM_ERROR;
    // under tensor parallelism the context carries the PER-RANK head counts, like the attention operator
    hidden_ = num_heads_ * size_per_head_;
    kv_stride_ = group_num_ * size_per_head_;
    // inv_freq exactly as RotaryOp::calculate_invfreq (rotary_op.h:51-76, base_rotary): float pow of a float exponent
    std::vector<float> inv(size_per_head_ / 2);
    for (int i = 0; i < size_per_head_ / 2; ++i) inv[i] = 1.f / std::pow(base_, float(i * 2) / float(size_per_head_));
    inv_freq_ = std::make_unique<AsTensor>(op_name_ + ".inv_freq", DeviceType::HIP, FLOAT32, Shape{(int64_t)inv.size()});
    if (!inv_freq_->GetDataPtr()) return AsStatus::ALLSPARK_MEMORY_ERROR;
    if (hipMemcpy(inv_freq_->GetDataPtr(), inv.data(), inv.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
      return AsStatus::ALLSPARK_RUNTIME_ERROR;
    positions_ = std::make_unique<AsTensor>(op_name_ + ".positions", DeviceType::HIP, INT32, Shape{1});
    return AsStatus::ALLSPARK_SUCCESS;
  }
  AsStatus Reshape(RuntimeContext*) override {
    AsTensor* x = tensor_map_->at(in_names_[0]).get();
    Shape s = x->GetShape();
    if (s.size() != 3 || (int)s[2] != hidden_ + 2 * kv_stride_) return AsStatus::ALLSPARK_RUNTIME_ERROR;  // rotary_op.cpp:259-266
    batch_ = (int)s[0];
    seq_len_ = (int)s[1];
    AsTensor* y = tensor_map_->at(out_names_[0]).get();
    y->SetDataType(x->GetDataType());
    AS_CHECK_STATUS(y->SetShape(std::move(s)));
    return positions_->SetShape(Shape{(int64_t)std::max(1, batch_ * seq_len_)});
  }
  AsStatus Forward(RuntimeContext* rt) override {
    AsTensor* x = tensor_map_->at(in_names_[0]).get();
    AsTensor* y = tensor_map_->at(out_names_[0]).get();
    const int rows = batch_ * seq_len_;
    if (rows == 0) return AsStatus::ALLSPARK_SUCCESS;
    // positions staged in pinned memory, guarded by an event (a pageable copy + hipStreamSynchronize per layer per step
    // serialised the stream); the reference stages them once per step through its layer cache manager (rotary_op.cpp:315-338)
    if ((size_t)rows > pos_cap_) {
      if (pos_host_) (void)hipHostFree(pos_host_);
      pos_cap_ = std::max<size_t>(rows, 256);
      if (hipHostMalloc((void**)&pos_host_, pos_cap_ * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) return AsStatus::ALLSPARK_MEMORY_ERROR;
    }
    if (!staged_ && hipEventCreateWithFlags(&staged_, hipEventDisableTiming) != hipSuccess) return AsStatus::ALLSPARK_RUNTIME_ERROR;
    if (hipEventSynchronize(staged_) != hipSuccess) return AsStatus::ALLSPARK_RUNTIME_ERROR;  // the previous copy has left the buffer
    uint32_t* pos = pos_host_;
    if (rt->is_context) {
      if (batch_ != 1) return AsStatus::ALLSPARK_RUNTIME_ERROR;  // rotary_op.cpp:301-306
      const GenerateContext* g = rt->GetContextGenCtx();
      // model.cpp:532 sets gen_ctx->step = prefix_len for a prefill over a cached prefix, and the kernel rotates row t at
      // seq_pos + step (rotary.cu:133): the prefix enters ONCE, through step (step + prefix_len counted it twice)
      for (int t = 0; t < seq_len_; ++t) pos[t] = (uint32_t)(g->step + t);
    } else {
      if (rt->GetGenCtxListSize() != batch_ || seq_len_ != 1) return AsStatus::ALLSPARK_RUNTIME_ERROR;
      for (int b = 0; b < batch_; ++b) pos[b] = (uint32_t)rt->GetGenCtx(b)->step;  // rotary_op.cpp:379-389
    }
    hipStream_t s = stream_of(ctx_);
    if (hipMemcpyAsync(positions_->GetDataPtr(), pos, rows * sizeof(uint32_t), hipMemcpyHostToDevice, s) != hipSuccess ||
        hipEventRecord(staged_, s) != hipSuccess)
      return AsStatus::ALLSPARK_RUNTIME_ERROR;
    if (y->GetDataPtr() != x->GetDataPtr() &&
        hipMemcpyAsync(y->GetDataPtr(), x->GetDataPtr(), x->GetSizeInByte(), hipMemcpyDeviceToDevice, s) != hipSuccess)
      return AsStatus::ALLSPARK_RUNTIME_ERROR;
    return FromDihip(dihip_rope_qk(s, y->GetDataPtr(), (const uint32_t*)positions_->GetDataPtr(), (const float*)inv_freq_->GetDataPtr(),
                                   rows, num_heads_, group_num_, size_per_head_, DihipDtype(x->GetDataType())));
  }

 private:
  int num_heads_ = 0, group_num_ = 0, size_per_head_ = 0, hidden_ = 0, kv_stride_ = 0, batch_ = 0, seq_len_ = 0;
  float base_ = 10000.f;
  std::unique_ptr<AsTensor> inv_freq_, positions_;
  uint32_t* pos_host_ = nullptr;
  size_t pos_cap_ = 0;
  hipEvent_t staged_ = nullptr;
};
REGISTER_OP(Rotary, HIP, RotaryHIP)

// ------------------------------------------------------------------------------------------------ Binary
class BinaryHIP : public AsOperator {
 public:
  explicit BinaryHIP(const std::string& t = "") : AsOperator(t) {}
  AsStatus Init(const OperatorProto& op_proto, const DeviceContext& ctx, const TensorMap& weights_map, TensorMap* tensor_map) override {
    AS_CHECK_STATUS(AsOperator::Init(op_proto, ctx, weights_map, tensor_map));
    if (in_names_.size() != 2) return AsStatus::ALLSPARK_PARAM_ERROR;
    const char* p = attr_ptr(op_proto, "binary_type");
    if (!p) return AsStatus::ALLSPARK_PARAM_ERROR;  // binary_op.cpp:28-31
    type_ = *(const int*)p;
    if (type_ != 1 && type_ != 2) return AsStatus::ALLSPARK_PARAM_ERROR;  // ADD, MUL (GEGLU / SWIGLU: see UnaryGLU)
    tensor_map_->at(out_names_[0])->SetDataType(tensor_map_->at(in_names_[0])->GetDataType());
    return AsStatus::ALLSPARK_SUCCESS;
  }
  AsStatus Reshape(RuntimeContext*) override {
    AsTensor* a = tensor_map_->at(in_names_[0]).get();
    AsTensor* y = tensor_map_->at(out_names_[0]).get();
    y->SetDataType(a->GetDataType());
    Shape s = a->GetShape();
    return y->SetShape(std::move(s));
  }
  AsStatus Forward(RuntimeContext*) override {
    AsTensor* a = tensor_map_->at(in_names_[0]).get();
    AsTensor* b = tensor_map_->at(in_names_[1]).get();
    AsTensor* y = tensor_map_->at(out_names_[0]).get();
    if (a->GetDataType() != b->GetDataType() || a->Count() != b->Count() || !is_ft(a->GetDataType())) return AsStatus::ALLSPARK_PARAM_ERROR;
    auto fn = type_ == 1 ? dihip_binary_add : dihip_binary_mul;
    return FromDihip(fn(stream_of(ctx_), y->GetDataPtr(), a->GetDataPtr(), b->GetDataPtr(), (size_t)a->Count(), DihipDtype(a->GetDataType())));
  }

 private:
  int type_ = 0;
};
REGISTER_OP(Binary, HIP, BinaryHIP)

// ------------------------------------------------------------------------------------------------ Unary / UnaryGLU
template <bool GLU>
class UnaryHIPT : public AsOperator {
 public:
  explicit UnaryHIPT(const std::string& t = "") : AsOperator(t) {}
  AsStatus Init(const OperatorProto& op_proto, const DeviceContext& ctx, const TensorMap& weights_map, TensorMap* tensor_map) override {
    AS_CHECK_STATUS(AsOperator::Init(op_proto, ctx, weights_map, tensor_map));
    const char* p = attr_ptr(op_proto, "unary_type");
    if (!p) return AsStatus::ALLSPARK_PARAM_ERROR;  // unary_op.cpp:25-28
    type_ = *(const int*)p;
    if (type_ < TANH || type_ > SIGMOID) return AsStatus::ALLSPARK_PARAM_ERROR;
    tensor_map_->at(out_names_[0])->SetDataType(tensor_map_->at(in_names_[0])->GetDataType());
    return AsStatus::ALLSPARK_SUCCESS;
  }
  AsStatus Reshape() override {
    AsTensor* x = tensor_map_->at(in_names_[0]).get();
    AsTensor* y = tensor_map_->at(out_names_[0]).get();
    y->SetDataType(x->GetDataType());
    Shape s = x->GetShape();
    if (GLU) {
      if (s.empty() || s.back() % 2) return AsStatus::ALLSPARK_RUNTIME_ERROR;  // unary_glu_op.cpp:61-65
      inner_ = (size_t)s.back() / 2;
      outer_ = (size_t)x->Count() / (size_t)s.back();
      s.back() = (int64_t)inner_;
    }
    return y->SetShape(std::move(s));
  }
  AsStatus Forward() override {
    AsTensor* x = tensor_map_->at(in_names_[0]).get();
    AsTensor* y = tensor_map_->at(out_names_[0]).get();
    if (!is_ft(x->GetDataType())) return AsStatus::ALLSPARK_PARAM_ERROR;
    if (GLU)
      return FromDihip(dihip_unary_glu(stream_of(ctx_), y->GetDataPtr(), x->GetDataPtr(), outer_, inner_, type_, DihipDtype(x->GetDataType())));
    return FromDihip(dihip_unary(stream_of(ctx_), y->GetDataPtr(), x->GetDataPtr(), (size_t)x->Count(), type_, DihipDtype(x->GetDataType())));
  }

 private:
  int type_ = 0;
  size_t outer_ = 0, inner_ = 0;
};
using UnaryHIP = UnaryHIPT<false>;
using UnaryGLUHIP = UnaryHIPT<true>;
REGISTER_OP(Unary, HIP, UnaryHIP)
REGISTER_OP(UnaryGLU, HIP, UnaryGLUHIP)

// ------------------------------------------------------------------------------------------------ EmbeddingT5
class EmbeddingT5HIP : public AsOperator {
 public:
  explicit EmbeddingT5HIP(const std::string& t = "") : AsOperator(t) {}
  AsStatus Init(const OperatorProto& op_proto, const DeviceContext& ctx, const TensorMap& weights_map, TensorMap* tensor_map) override {
    AS_CHECK_STATUS(AsOperator::Init(op_proto, ctx, weights_map, tensor_map));
    if (weights_.size() != 1 || weights_[0]->GetShape().size() != 2) return AsStatus::ALLSPARK_PARAM_ERROR;  // token-type table: not in this graph
    vocab_ = (int)weights_[0]->GetShape()[0];
    hidden_ = (int)weights_[0]->GetShape()[1];
    if (!is_ft(weights_[0]->GetDataType())) return AsStatus::ALLSPARK_PARAM_ERROR;
    tensor_map_->at(out_names_[0])->SetDataType(weights_[0]->GetDataType());
    return AsStatus::ALLSPARK_SUCCESS;
  }
  AsStatus Reshape(RuntimeContext*) override {
    AsTensor* ids = tensor_map_->at(in_names_[0]).get();
    if (ids->GetDataType() != INT64 || ids->GetShape().size() != 2) return AsStatus::ALLSPARK_PARAM_ERROR;
    batch_ = (int)ids->GetShape()[0];
    seq_ = (int)ids->GetShape()[1];
    return tensor_map_->at(out_names_[0])->SetShape(Shape{batch_, seq_, hidden_});  // embeddingT5_op.cpp Reshape
  }
  AsStatus Forward(RuntimeContext*) override {
    AsTensor* ids = tensor_map_->at(in_names_[0]).get();
    AsTensor* y = tensor_map_->at(out_names_[0]).get();
    return FromDihip(dihip_embedding_ft(stream_of(ctx_), y->GetDataPtr(), (const int64_t*)ids->GetDataPtr(), weights_[0]->GetDataPtr(),
                                        batch_ * seq_, hidden_, vocab_, DihipDtype(weights_[0]->GetDataType())));
  }

 private:
  int vocab_ = 0, hidden_ = 0, batch_ = 0, seq_ = 0;
};
REGISTER_OP(EmbeddingT5, HIP, EmbeddingT5HIP)

// ------------------------------------------------------------------------------------------------ GetLastLine
class GetLastLineHIP : public AsOperator {
 public:
  explicit GetLastLineHIP(const std::string& t = "") : AsOperator(t) {}
  AsStatus Init(const OperatorProto& op_proto, const DeviceContext& ctx, const TensorMap& weights_map, TensorMap* tensor_map) override {
    AS_CHECK_STATUS(AsOperator::Init(op_proto, ctx, weights_map, tensor_map));
    tensor_map_->at(out_names_[0])->SetDataType(tensor_map_->at(in_names_[0])->GetDataType());
    return AsStatus::ALLSPARK_SUCCESS;
  }
  AsStatus Reshape(RuntimeContext*) override {
    AsTensor* x = tensor_map_->at(in_names_[0]).get();
    if (x->GetShape().size() != 3) return AsStatus::ALLSPARK_PARAM_ERROR;
    batch_ = (int)x->GetShape()[0];
    seq_ = (int)x->GetShape()[1];
    hidden_ = (int)x->GetShape()[2];
    AsTensor* y = tensor_map_->at(out_names_[0]).get();
    y->SetDataType(x->GetDataType());
    AS_CHECK_STATUS(y->SetShape(Shape{ctx_->GetModelMaxBatch(), 1, hidden_}));  // (warm-up size first, get_last_line.cpp:33-37)
    return y->SetShape(Shape{batch_, 1, hidden_});
  }
  AsStatus Forward(RuntimeContext*) override {
    AsTensor* x = tensor_map_->at(in_names_[0]).get();
    AsTensor* y = tensor_map_->at(out_names_[0]).get();
    const size_t es = SizeofType(x->GetDataType());
    // as the reference (get_last_line.cpp:81-87): batch * hidden elements from row seq - 1 on -- the context phase runs
    // one request (batch 1), the decoder phase has seq = 1
    const hipError_t e = hipMemcpyAsync(y->GetDataPtr(), (const char*)x->GetDataPtr() + (size_t)(seq_ - 1) * hidden_ * es,
                                        (size_t)batch_ * hidden_ * es, hipMemcpyDeviceToDevice, stream_of(ctx_));
    return e == hipSuccess ? AsStatus::ALLSPARK_SUCCESS : AsStatus::ALLSPARK_RUNTIME_ERROR;
  }

 private:
  int batch_ = 0, seq_ = 0, hidden_ = 0;
};
REGISTER_OP(GetLastLine, HIP, GetLastLineHIP)

// ------------------------------------------------------------------------------------------------ GenerateOp (greedy)
class GenerateOpHIP : public AsOperator {
 public:
  explicit GenerateOpHIP(const std::string& t = "") : AsOperator(t) {}
  AsStatus Init(const OperatorProto& op_proto, const DeviceContext& ctx, const TensorMap& weights_map, TensorMap* tensor_map) override {
    AS_CHECK_STATUS(AsOperator::Init(op_proto, ctx, weights_map, tensor_map));
    // (the graph's own top_k attribute is a default the per-request gen_cfg overrides: generate_op.cpp:336-343)
    tensor_map_->at(out_names_[0])->SetDataType(INT64);
    return AsStatus::ALLSPARK_SUCCESS;
  }
  AsStatus Reshape(RuntimeContext* rt) override {
    AsTensor* x = tensor_map_->at(in_names_[0]).get();
    const Shape& s = x->GetShape();
    if (s.size() < 2) return AsStatus::ALLSPARK_PARAM_ERROR;
    vocab_ = (int)s.back();
    rows_ = (int)(x->Count() / vocab_);
    // GenerateOp::Reshape: [batch, seq, vocab], the last row of each request is sampled (generate_op.cpp:326-332,476-477)
    seq_ = s.size() >= 3 ? (int)s[s.size() - 2] : 1;
    rows_ = (int)(x->Count() / vocab_) / std::max(seq_, 1);
    AsTensor* y = tensor_map_->at(out_names_[0]).get();
    y->SetDataType(INT64);
    AS_CHECK_STATUS(y->SetShape(Shape{rows_, 1}));
    if (rt && rt->GetGenCtxListSize() > 0) {
      AS_CHECK_STATUS(params_.Gather(rt, rows_, stream_of(ctx_)));
      // the logits processors before sampling and the log-probabilities after it (generate_op.cpp:536-538, :600-606): staged form
      std::string why;
      const AsStatus st = proc_.Gather(rt, rows_, vocab_, ctx_->GetModelMaxLength(), false, stream_of(ctx_), &why);
      if (st != AsStatus::ALLSPARK_SUCCESS) {
        std::fprintf(stderr, "[dashinfer_hip] GenerateOp: %s\n", why.c_str());
        return st;
      }
    }
    // scratch: f32 copy of FT logits + the arg-max partials (64 pairs per row)
    const int64_t need = (x->GetDataType() == FLOAT32 ? 0 : (int64_t)rows_ * vocab_ * 4) + (int64_t)rows_ * 64 * 8 + 256;
    AsTensor* wsp = tensor_map_->at("workspace").get();
    if (wsp->GetSizeInByte() < (size_t)need) AS_CHECK_STATUS(wsp->SetShape(Shape{need}));
    return AsStatus::ALLSPARK_SUCCESS;
  }
  AsStatus Forward(RuntimeContext* rt) override {
    AsTensor* x = tensor_map_->at(in_names_[0]).get();
    AsTensor* y = tensor_map_->at(out_names_[0]).get();
    AsTensor* wsp = tensor_map_->at("workspace").get();
    hipStream_t s = stream_of(ctx_);
    char* ws = (char*)wsp->GetDataPtr();
    if (seq_ > 1 && rows_ != 1) return AsStatus::ALLSPARK_PARAM_ERROR;  // several rows per request only in the context phase (one request)
    const size_t es = SizeofType(x->GetDataType());
    const char* last = (const char*)x->GetDataPtr() + (size_t)(std::max(seq_, 1) - 1) * vocab_ * es;  // in_ptr, generate_op.cpp:476-477
    const float* logits = (const float*)last;
    size_t off = 0;
    if (x->GetDataType() != FLOAT32) {
      if (!is_ft(x->GetDataType())) return AsStatus::ALLSPARK_PARAM_ERROR;
      AS_CHECK_STATUS(FromDihip(dihip_cast_to_f32(s, (float*)ws, last, (size_t)rows_ * vocab_, DihipDtype(x->GetDataType()))));
      logits = (const float*)ws;
      off = ((size_t)rows_ * vocab_ * 4 + 255) & ~(size_t)255;
    }
    const bool have_rt = rt && rt->GetGenCtxListSize() > 0;
    if (have_rt && proc_.any_processors()) {  // in place on the scores, as the reference does (generate_impl_gpu.hpp:105-109)
      std::string why;
      const AsStatus st = proc_.StageHistory(rt, s, &why);
      if (st != AsStatus::ALLSPARK_SUCCESS) {
        std::fprintf(stderr, "[dashinfer_hip] GenerateOp: %s\n", why.c_str());
        return st;
      }
      AS_CHECK_STATUS(proc_.RunStaged(const_cast<float*>(logits), s));
    }
    if (have_rt && params_.any_sampling()) {
      AS_CHECK_STATUS(params_.StagePositions(rt, rt->is_context ? seq_ : 1, s));
      AS_CHECK_STATUS(FromDihip(dihip_sample_rows(s, (int64_t*)y->GetDataPtr(), logits, rows_, vocab_, params_.top_k(), params_.top_p(),
                                                  params_.temperature(), params_.seed(), params_.dev_pos(), nullptr, nullptr, params_.wide_rows())));
    } else {
      AS_CHECK_STATUS(FromDihip(dihip_argmax(s, (int64_t*)y->GetDataPtr(), logits, rows_, vocab_, ws + off, wsp->GetSizeInByte() - off)));
    }
    if (have_rt && proc_.any_logprobs()) return proc_.LogprobsStaged(rt, logits, (const int64_t*)y->GetDataPtr(), ctx_->GetRank() == 0, s);
    return AsStatus::ALLSPARK_SUCCESS;
  }

 private:
  int rows_ = 0, vocab_ = 0, seq_ = 1;
  SamplingParams params_;
  LogitsProcParams proc_;
};
REGISTER_OP(GenerateOp, HIP, GenerateOpHIP)

}  // namespace allspark
