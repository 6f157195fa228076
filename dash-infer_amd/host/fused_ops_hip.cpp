// fused_ops_hip.cpp -- the FUSED decode-step operators of the HIP backend, behind the allspark operator interface.
//
// The reference graph runs a Qwen2 layer as fourteen operators (python/pyhie/allspark/model/qwen_v15.py:210-381):
//   LayerNormNoBeta -> GemmA16Wx(qkv) -> Rotary -> DecOptMQA -> GemmA16Wx(o) -> Binary ADD -> LayerNormNoBeta ->
//   GemmA16Wx(gate, SILU) | GemmA16Wx(up) -> Binary MUL -> GemmA16Wx(down) -> Binary ADD
// which on this backend is fourteen launches per layer.  The device library's decode-step entry points
// (include/dashinfer_hip.h sections 1 and 3b) do the same arithmetic in five: the norms ride in GEMV prologues, the residual
// adds / SwiGLU in epilogues, Rotary + DecoderCacheAppend + SpanAttention are one launch, and the residual stream stays f32
// (the x86 reference's activation type, gemm_op_cpu.cpp:75-126).  host/fusion_pass.cpp rewrites an OperatorProto list of the
// reference graph into these operator types; they are ordinary AsOperators (Init / Reshape / Alloc / Forward,
// csrc/core/operator/operator.h:38-201) registered with REGISTER_OP for DeviceType::HIP and call the SAME C-ABI entries,
// with the same arguments, as decoder.DecodeSession.step (dash-infer_amd/decoder.py) -- the logits are bit-identical
// (tests/test_gpu_host_runner.py).
//
//   DihipEmbedding      <- EmbeddingT5                                    f32 hidden rows
//   DihipNormGemm       <- LayerNormNoBeta + GemmA16W8|W4 (qkv, bias)     [4 < M <= 32: takes the norm its producer made]
//   DihipRopeSpanAttn   <- Rotary + DecOptMQA|DecOptMHA                   (class derived from SpanAttnOpHIP)
//   DihipGemmAddTo      <- GemmA16Wx + [AllReduce] + Binary ADD           [4 < M <= 32: + the LayerNormNoBeta that follows]
//   DihipNormSwiGLU     <- LayerNormNoBeta + GemmA16Wx(SILU) + GemmA16Wx + Binary MUL
//   DihipMoeBlock       <- the mixture-of-experts feed-forward block of qwen_v20_moe.py:318-382 (twelve operators)
//   DihipLMHead         <- LayerNormNoBeta + GetLastLine + Gemm(lm_head)  f32 logits
//   DihipGreedy         <- GenerateOp (greedy requests); advances the device-resident length counters in the same launch
//
// Activation layouts between two fused operators (row-major / DIHIP_ACT_FRAG32) are negotiated through HIPContext: the consumer
// advertises its GEMM shape for the tensor at Init, the producer decides at Reshape when the row count is known
// (decoder.DecodeSession: ops.prefers_frag).
#include <cstdio>
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "dashinfer_hip.h"
#include "moe_op_hip.h"
#include "operator.h"
#include "sampling_host.h"
#include "span_attn_op_hip.h"

namespace allspark {

namespace {

const HIPContext& hip_ctx(const DeviceContext* ctx) { return *static_cast<const HIPContext*>(ctx); }
hipStream_t stream_of(const DeviceContext* ctx) { return hip_ctx(ctx).GetStream(); }
const char* attr_ptr(const OperatorProto& p, const char* k) {
  auto it = p.attr.find(k);
  return it == p.attr.end() ? nullptr : it->second.c_str();
}
bool env_on(const char* name, bool dflt) {
  const char* e = getenv(name);
  return e ? e[0] != '0' : dflt;
}

// DecodeSession.norm_fuse (decoder.py): batched decode, one rank -- the RMSNorm after a residual GEMM is produced by that GEMM
bool norm_fuse_active(const DeviceContext* ctx, const RuntimeContext* rt, int m) {
  static const bool enabled = env_on("DIHIP_DECODER_NORM_FUSE", true);
  return enabled && rt && !rt->is_context && m > 4 && m <= 32 && ctx->GetNranks() <= 1;
}

// grows `t` so that it can hold `bytes` (FRAG32 buffers are padded to 16 / 32 rows), keeps its logical shape, and zero-fills
// storage that was (re)allocated: the fragment layout's padding rows are read by the matrix cores
AsStatus ensure_capacity_zeroed(AsTensor* t, size_t bytes, Shape logical, hipStream_t s) {
  void* before = t->GetDataPtr();
  const size_t es = SizeofType(t->GetDataType());
  AS_CHECK_STATUS(t->SetShape(Shape{(int64_t)((bytes + es - 1) / es)}));
  void* after = t->GetDataPtr();
  if (after != before && after != nullptr && hipMemsetAsync(after, 0, bytes, s) != hipSuccess) return AsStatus::ALLSPARK_RUNTIME_ERROR;
  return t->SetShape(std::move(logical));
}

AsStatus grow_workspace(TensorMap* tm, size_t bytes) {
  AsTensor* wsp = tm->at("workspace").get();  // shared scratch, grows only (model.cpp:241-243)
  if (wsp->GetSizeInByte() < bytes) return wsp->SetShape(Shape{(int64_t)bytes});
  return AsStatus::ALLSPARK_SUCCESS;
}

// a quantised weight in dihip tile-major order, re-laid-out once at Init (GemmA16W8GPU::InitV2 does the same job for the CUDA
// kernels, gemm_a16w8_gpu.cpp:421-473)
struct PackedLowp {
  int wbits = 0, n = 0, k = 0, group = -1;
  DataType ft = BFLOAT16;
  std::unique_ptr<AsTensor> w, sz;
  AsStatus Pack(const std::string& name, int bits, int group_size, const AsTensor* wq, const AsTensor* scales, const AsTensor* zeros,
                hipStream_t s, bool release_source = true) {
    wbits = bits;
    group = group_size;
    if (wbits != 4 && wbits != 8) return AsStatus::ALLSPARK_PARAM_ERROR;
    if (group != -1 && (group < 32 || group % 32 != 0)) return AsStatus::ALLSPARK_PARAM_ERROR;
    if (wq->GetShape().size() != 2) return AsStatus::ALLSPARK_PARAM_ERROR;
    k = (int)wq->GetShape()[0];
    n = (int)scales->GetShape().back();
    if (wbits == 8 && (wq->GetDataType() != INT8 || (int)wq->GetShape()[1] != n)) return AsStatus::ALLSPARK_PARAM_ERROR;
    if (wbits == 4 && (wq->GetDataType() != UINT8 || (int)wq->GetShape()[1] != (n + 1) / 2)) return AsStatus::ALLSPARK_PARAM_ERROR;
    ft = scales->GetDataType();
    if (ft != FLOAT16 && ft != BFLOAT16) return AsStatus::ALLSPARK_PARAM_ERROR;
    w = std::make_unique<AsTensor>(name + ".packed_w", DeviceType::HIP, INT8, Shape{(int64_t)dihip_gemm_lowp_packed_weight_bytes(wbits, n, k)});
    sz = std::make_unique<AsTensor>(name + ".packed_sz", DeviceType::HIP, INT8, Shape{(int64_t)dihip_gemm_lowp_packed_sz_bytes(n, k, group)});
    if (!w->GetDataPtr() || !sz->GetDataPtr()) return AsStatus::ALLSPARK_MEMORY_ERROR;
    AS_CHECK_STATUS(FromDihip(dihip_gemm_lowp_pack(s, wbits, wq->GetDataPtr(), scales->GetDataPtr(), zeros->GetDataPtr(), n, k, group,
                                                   DihipDtype(ft), w->GetDataPtr(), sz->GetDataPtr())));
    // The reference re-lays-out IN PLACE (gemm_a16w8_gpu.cpp:456-469: weights_buffer synced, no second copy).  The tile-major form
    // has another size, so the source is released instead once the pack launch has read it -- when the weight map owns it (a view
    // of caller memory, as the test harness registers, is the caller's to free): no weight is held twice.
    if (release_source && wq->OwnsStorage()) {  // (release_source = false: a scratch tensor the caller re-uses)
      if (hipStreamSynchronize(s) != hipSuccess) return AsStatus::ALLSPARK_RUNTIME_ERROR;
      const_cast<AsTensor*>(wq)->Free();
    }
    return AsStatus::ALLSPARK_SUCCESS;
  }
  ActLayoutPref pref(int dual) const { return ActLayoutPref{wbits, n, k, group, dual, ft == BFLOAT16 ? 1 : 0}; }
  // (the small-batch kernels, and with them the FRAG32 layout, serve bf16 and -- since round 5 -- f16)
  bool prefers_frag(int m, int dual) const { return dihip_gemm_lowp_prefers_frag(wbits, m, n, k, group, dual) != 0; }
};
bool pref_frag(const ActLayoutPref* p, int m) {
  return p && dihip_gemm_lowp_prefers_frag(p->wbits, m, p->n, p->k, p->group, p->dual) != 0;
}

std::unique_ptr<AsTensor> zeroed(const std::string& name, size_t bytes, hipStream_t s) {
  auto t = std::make_unique<AsTensor>(name, DeviceType::HIP, INT8, Shape{(int64_t)bytes});
  if (t->GetDataPtr() && hipMemsetAsync(t->GetDataPtr(), 0, bytes, s) != hipSuccess) t.reset();
  return t;
}

int read_wbits(const OperatorProto& p) {
  const char* a = attr_ptr(p, "wbits");
  return a ? *(const int*)a : 0;
}
int read_group(const OperatorProto& p) {
  const char* a = attr_ptr(p, "GroupSize");
  return a ? *(const int*)a : -1;
}

// ---- the attention half of a batch-1 decode layer as ONE launch (dihip_decode_attn_block, round 5) ---------------------------------
// DihipNormGemm(qkv) -> DihipRopeSpanAttn -> DihipGemmAddTo(o) stay three operators of the list (context phase, batches, quantised
// caches run them as before); for ONE request on the 16-bit cache with int4 g128 (or int8 per-channel) weights the o-projection operator, which finds
// the two in front of it through HIPContext::Producer at Init, tells them at Reshape to skip their launches and issues the
// single launch in its own Forward with what they hand over here.
struct AttnBlockQkv {
  const AsTensor* h = nullptr;  // f32 hidden rows (input of the norm)
  const void* gamma = nullptr;
  float eps = 0.f;
  const PackedLowp* w = nullptr;
  const void* bias = nullptr;
  int act = 0, m = 0;
};
class AttnBlockQkvPart {
 public:
  virtual ~AttnBlockQkvPart() = default;
  virtual AttnBlockQkv BlockQkv() const = 0;
  virtual void BlockSkip(bool skip) = 0;
};
struct AttnBlockAttn {
  void* const* kd = nullptr;
  void* const* vd = nullptr;
  const uint32_t* old_lens = nullptr;
  const float* rope_tab = nullptr;
  AsTensor* ws = nullptr;
  int n = 0, g = 0, h = 0, span = 0, max_spans = 0, kv_mode = 0, batch = 0, seq = 0;
  DataType dt = BFLOAT16;
  float alpha = 0.f;
  std::string qkv_name;  // its input: the qkv operator's output
};
class AttnBlockAttnPart {
 public:
  virtual ~AttnBlockAttnPart() = default;
  virtual AttnBlockAttn BlockAttn() const = 0;  // valid after this step's Forward (the lengths are staged there)
  virtual void BlockSkip(bool skip) = 0;
};

}  // namespace

// ===================================================================================================== DihipEmbedding
// EmbeddingT5 into the f32 hidden stream: h[m, :] = float(table[ids[m], :]), ids clamped to the vocabulary.
class DihipEmbeddingOp : public AsOperator {
 public:
  explicit DihipEmbeddingOp(const std::string& t = "") : AsOperator(t) {}
  AsStatus Init(const OperatorProto& op_proto, const DeviceContext& ctx, const TensorMap& weights_map, TensorMap* tensor_map) override {
    AS_CHECK_STATUS(AsOperator::Init(op_proto, ctx, weights_map, tensor_map));
    if (weights_.size() != 1 || weights_[0]->GetShape().size() != 2) return AsStatus::ALLSPARK_PARAM_ERROR;
    vocab_ = (int)weights_[0]->GetShape()[0];
    hidden_ = (int)weights_[0]->GetShape()[1];
    if (weights_[0]->GetDataType() != BFLOAT16 && weights_[0]->GetDataType() != FLOAT16) return AsStatus::ALLSPARK_PARAM_ERROR;
    tensor_map_->at(out_names_[0])->SetDataType(FLOAT32);
    return AsStatus::ALLSPARK_SUCCESS;
  }
  AsStatus Reshape(RuntimeContext*) override {
    AsTensor* ids = tensor_map_->at(in_names_[0]).get();
    if (ids->GetDataType() != INT64 || ids->GetShape().size() != 2) return AsStatus::ALLSPARK_PARAM_ERROR;
    batch_ = (int)ids->GetShape()[0];
    seq_ = (int)ids->GetShape()[1];
    AsTensor* y = tensor_map_->at(out_names_[0]).get();
    y->SetDataType(FLOAT32);
    return y->SetShape(Shape{batch_, seq_, hidden_});
  }
  AsStatus Forward(RuntimeContext*) override {
    AsTensor* ids = tensor_map_->at(in_names_[0]).get();
    AsTensor* y = tensor_map_->at(out_names_[0]).get();
    return FromDihip(dihip_embedding_v(stream_of(ctx_), (float*)y->GetDataPtr(), (const int64_t*)ids->GetDataPtr(), weights_[0]->GetDataPtr(),
                                       batch_ * seq_, hidden_, vocab_, DihipDtype(weights_[0]->GetDataType())));
  }

 private:
  int vocab_ = 0, hidden_ = 0, batch_ = 0, seq_ = 0;
};
REGISTER_OP(DihipEmbedding, HIP, DihipEmbeddingOp)

// ====================================================================================================== DihipNormGemm
// y = act(RMSNorm(h; gamma, eps) . W + bias): weights [gamma, W, scales, zeros, (bias)]; inputs [h, (xnorm)]; attrs eps, wbits,
// GroupSize, activation.  With a second input and 4 < M <= 32 in the decoder phase the normalised rows come from the producer
// of h (DihipGemmAddTo) and the GEMM reads them directly (dihip_prenorm_gemm).
class DihipNormGemmOp : public AsOperator, public AttnBlockQkvPart {
 public:
  explicit DihipNormGemmOp(const std::string& t = "") : AsOperator(t) {}
  AsStatus Init(const OperatorProto& op_proto, const DeviceContext& ctx, const TensorMap& weights_map, TensorMap* tensor_map) override {
    AS_CHECK_STATUS(AsOperator::Init(op_proto, ctx, weights_map, tensor_map));
    if (ctx.GetDeviceType() != DeviceType::HIP) return AsStatus::ALLSPARK_PARAM_ERROR;
    if (weights_.size() != 4 && weights_.size() != 5) return AsStatus::ALLSPARK_PARAM_ERROR;
    const char* e = attr_ptr(op_proto, "eps");
    if (!e) return AsStatus::ALLSPARK_PARAM_ERROR;  // layernorm_nobeta_op.cpp:69-73
    eps_ = *(const float*)e;
    if (const char* a = attr_ptr(op_proto, "activation")) act_ = *(const int*)a;
    AS_CHECK_STATUS(w_.Pack(op_name_, read_wbits(op_proto), read_group(op_proto), weights_[1], weights_[2], weights_[3], stream_of(&ctx)));
    if ((int)weights_[0]->GetShape()[0] != w_.k || weights_[0]->GetDataType() != w_.ft) return AsStatus::ALLSPARK_PARAM_ERROR;
    sync_ = zeroed(op_name_ + ".sync", dihip_gemm_lowp_sync_bytes(), stream_of(&ctx));
    if (!sync_) return AsStatus::ALLSPARK_MEMORY_ERROR;
    if (in_names_.size() > 1) hip_ctx(&ctx).AdvertiseLayoutPref(in_names_[1], w_.pref(0));
    hip_ctx(&ctx).RegisterConsumerWeights(in_names_[0], w_.w->GetDataPtr(), w_.w->GetSizeInByte());
    tensor_map_->at(out_names_[0])->SetDataType(w_.ft);
    hip_ctx(&ctx).RegisterProducer(out_names_[0] + "\x01qkv", static_cast<AttnBlockQkvPart*>(this));
    return AsStatus::ALLSPARK_SUCCESS;
  }
  AttnBlockQkv BlockQkv() const override {
    AttnBlockQkv q;
    q.h = tensor_map_->at(in_names_[0]).get();
    q.gamma = weights_[0]->GetDataPtr();
    q.eps = eps_;
    q.w = &w_;
    q.bias = weights_.size() == 5 ? weights_[4]->GetDataPtr() : nullptr;
    q.act = act_;
    q.m = m_;
    return q;
  }
  void BlockSkip(bool skip) override { skip_ = skip; }
  AsStatus Reshape(RuntimeContext*) override {
    skip_ = false;  // (the o-projection's Reshape, later in the list, decides anew)
    AsTensor* h = tensor_map_->at(in_names_[0]).get();
    Shape s = h->GetShape();
    if (s.empty() || (int)s.back() != w_.k || h->GetDataType() != FLOAT32) return AsStatus::ALLSPARK_PARAM_ERROR;
    m_ = (int)(h->Count() / w_.k);
    s.back() = w_.n;
    AsTensor* y = tensor_map_->at(out_names_[0]).get();
    y->SetDataType(w_.ft);
    AS_CHECK_STATUS(y->SetShape(std::move(s)));
    return grow_workspace(tensor_map_, dihip_gemm_lowp_workspace_bytes(w_.wbits, std::max(m_, 1), w_.n, w_.k, w_.group));
  }
  AsStatus Forward(RuntimeContext* rt) override {
    if (skip_) return AsStatus::ALLSPARK_SUCCESS;  // part of the o-projection's single launch (AttnBlockQkvPart)
    AsTensor* h = tensor_map_->at(in_names_[0]).get();
    AsTensor* y = tensor_map_->at(out_names_[0]).get();
    AsTensor* wsp = tensor_map_->at("workspace").get();
    const void* bias = weights_.size() == 5 ? weights_[4]->GetDataPtr() : nullptr;
    hipStream_t s = stream_of(ctx_);
    if (in_names_.size() > 1 && norm_fuse_active(ctx_, rt, m_)) {
      AsTensor* xn = tensor_map_->at(in_names_[1]).get();
      const RowNormState rn = hip_ctx(ctx_).RowNorm(in_names_[1]);  // (parts == 0: the finished norm)
      return FromDihip(dihip_prenorm_gemm_rowsq(s, w_.wbits, xn->GetDataPtr(), hip_ctx(ctx_).ActLayout(in_names_[1]), w_.w->GetDataPtr(),
                                                w_.sz->GetDataPtr(), bias, y->GetDataPtr(), m_, w_.n, w_.k, w_.group, act_, wsp->GetDataPtr(),
                                                wsp->GetSizeInByte(), sync_->GetDataPtr(), DihipDtype(w_.ft), rn.rowsq, rn.parts, rn.eps));
    }
    return FromDihip(dihip_fused_norm_gemm(s, w_.wbits, (const float*)h->GetDataPtr(), weights_[0]->GetDataPtr(), eps_, w_.w->GetDataPtr(),
                                           w_.sz->GetDataPtr(), bias, y->GetDataPtr(), m_, w_.n, w_.k, w_.group, act_, wsp->GetDataPtr(),
                                           wsp->GetSizeInByte(), sync_->GetDataPtr(), DihipDtype(w_.ft)));
  }

 private:
  PackedLowp w_;
  float eps_ = 1e-6f;
  int act_ = 0, m_ = 0;
  bool skip_ = false;
  std::unique_ptr<AsTensor> sync_;
};
REGISTER_OP(DihipNormGemm, HIP, DihipNormGemmOp)

// ==================================================================================================== DihipNormSwiGLU
// act = FT(SiLU(RMSNorm(h).Wgate)) * FT(RMSNorm(h).Wup): weights [gamma, Wg, Sg, Zg, Wu, Su, Zu]; inputs [h, (xnorm)].
// The output feeds the down projection only: when both run on the small-batch kernels it is written in FRAG32.
class DihipNormSwiGLUOp : public AsOperator {
 public:
  explicit DihipNormSwiGLUOp(const std::string& t = "") : AsOperator(t) {}
  AsStatus Init(const OperatorProto& op_proto, const DeviceContext& ctx, const TensorMap& weights_map, TensorMap* tensor_map) override {
    AS_CHECK_STATUS(AsOperator::Init(op_proto, ctx, weights_map, tensor_map));
    if (ctx.GetDeviceType() != DeviceType::HIP || weights_.size() != 7) return AsStatus::ALLSPARK_PARAM_ERROR;
    const char* e = attr_ptr(op_proto, "eps");
    if (!e) return AsStatus::ALLSPARK_PARAM_ERROR;
    eps_ = *(const float*)e;
    const int wb = read_wbits(op_proto), grp = read_group(op_proto);
    AS_CHECK_STATUS(g_.Pack(op_name_ + ".gate", wb, grp, weights_[1], weights_[2], weights_[3], stream_of(&ctx)));
    AS_CHECK_STATUS(u_.Pack(op_name_ + ".up", wb, grp, weights_[4], weights_[5], weights_[6], stream_of(&ctx)));
    if (g_.n != u_.n || g_.k != u_.k || g_.ft != u_.ft) return AsStatus::ALLSPARK_PARAM_ERROR;
    if ((int)weights_[0]->GetShape()[0] != g_.k || weights_[0]->GetDataType() != g_.ft) return AsStatus::ALLSPARK_PARAM_ERROR;
    sync_ = zeroed(op_name_ + ".sync", dihip_gemm_lowp_sync_bytes(), stream_of(&ctx));
    if (!sync_) return AsStatus::ALLSPARK_MEMORY_ERROR;
    if (in_names_.size() > 1) hip_ctx(&ctx).AdvertiseLayoutPref(in_names_[1], g_.pref(1));
    for (const PackedLowp* pw : {&g_, &u_}) hip_ctx(&ctx).RegisterConsumerWeights(in_names_[0], pw->w->GetDataPtr(), pw->w->GetSizeInByte());
    tensor_map_->at(out_names_[0])->SetDataType(g_.ft);
    return AsStatus::ALLSPARK_SUCCESS;
  }
  AsStatus Reshape(RuntimeContext* rt) override {
    AsTensor* h = tensor_map_->at(in_names_[0]).get();
    Shape s = h->GetShape();
    if (s.empty() || (int)s.back() != g_.k || h->GetDataType() != FLOAT32) return AsStatus::ALLSPARK_PARAM_ERROR;
    m_ = (int)(h->Count() / g_.k);
    s.back() = g_.n;
    AsTensor* y = tensor_map_->at(out_names_[0]).get();
    y->SetDataType(g_.ft);
    // DecodeSession.act_frag: both this pair and its consumer (the down projection) run on the small-batch kernels
    const bool frag = rt && !rt->is_context && g_.prefers_frag(m_, 1) && pref_frag(hip_ctx(ctx_).LayoutPref(out_names_[0]), m_);
    y_layout_ = frag ? DIHIP_ACT_FRAG32 : DIHIP_ACT_ROWMAJOR;
    hip_ctx(ctx_).SetActLayout(out_names_[0], y_layout_);
    const size_t bytes = frag ? dihip_act_frag_bytes(m_, g_.n) : (size_t)m_ * g_.n * SizeofType(g_.ft);
    AS_CHECK_STATUS(ensure_capacity_zeroed(y, bytes, std::move(s), stream_of(ctx_)));
    return grow_workspace(tensor_map_, dihip_gemm_lowp_workspace_bytes(g_.wbits, std::max(m_, 1), g_.n, g_.k, g_.group));
  }
  AsStatus Forward(RuntimeContext* rt) override {
    AsTensor* h = tensor_map_->at(in_names_[0]).get();
    AsTensor* y = tensor_map_->at(out_names_[0]).get();
    AsTensor* wsp = tensor_map_->at("workspace").get();
    hipStream_t s = stream_of(ctx_);
    if (in_names_.size() > 1 && norm_fuse_active(ctx_, rt, m_)) {
      AsTensor* xn = tensor_map_->at(in_names_[1]).get();
      const RowNormState rn = hip_ctx(ctx_).RowNorm(in_names_[1]);
      return FromDihip(dihip_prenorm_swiglu_rowsq(s, g_.wbits, xn->GetDataPtr(), hip_ctx(ctx_).ActLayout(in_names_[1]), g_.w->GetDataPtr(),
                                                  g_.sz->GetDataPtr(), u_.w->GetDataPtr(), u_.sz->GetDataPtr(), y->GetDataPtr(), m_, g_.n, g_.k,
                                                  g_.group, wsp->GetDataPtr(), wsp->GetSizeInByte(), sync_->GetDataPtr(), DihipDtype(g_.ft), y_layout_,
                                                  rn.rowsq, rn.parts, rn.eps));
    }
    return FromDihip(dihip_fused_norm_swiglu_ex(s, g_.wbits, (const float*)h->GetDataPtr(), weights_[0]->GetDataPtr(), eps_, g_.w->GetDataPtr(),
                                                g_.sz->GetDataPtr(), u_.w->GetDataPtr(), u_.sz->GetDataPtr(), y->GetDataPtr(), m_, g_.n, g_.k,
                                                g_.group, wsp->GetDataPtr(), wsp->GetSizeInByte(), sync_->GetDataPtr(), DihipDtype(g_.ft),
                                                y_layout_));
  }

 private:
  PackedLowp g_, u_;
  float eps_ = 1e-6f;
  int m_ = 0, y_layout_ = DIHIP_ACT_ROWMAJOR;
  std::unique_ptr<AsTensor> sync_;
};
REGISTER_OP(DihipNormSwiGLU, HIP, DihipNormSwiGLUOp)

// ===================================================================================================== DihipGemmAddTo
// h_out (f32) = h_res + x . W  -- the Gemm, the AllReduce's addend rule and the Binary ADD of the reference graph: under tensor
// parallelism only rank 0 adds the residual (gemm_op.cpp:133-137), the partial rows are summed by the AllReduce operator that
// stays in the graph behind this one.  weights [W, scales, zeros, (next_gamma)]; inputs [x, h_res]; outputs [h_out, (xnorm)].
// With next_gamma and 4 < M <= 32 in the decoder phase the LayerNormNoBeta that follows in the graph is produced here, riding on
// the split-K reduction (dihip_fused_gemm_addto_norm), in the layout its consumer advertised.
class DihipGemmAddToOp : public AsOperator {
 public:
  explicit DihipGemmAddToOp(const std::string& t = "") : AsOperator(t) {}
  AsStatus Init(const OperatorProto& op_proto, const DeviceContext& ctx, const TensorMap& weights_map, TensorMap* tensor_map) override {
    AS_CHECK_STATUS(AsOperator::Init(op_proto, ctx, weights_map, tensor_map));
    if (ctx.GetDeviceType() != DeviceType::HIP) return AsStatus::ALLSPARK_PARAM_ERROR;
    if (weights_.size() != 3 && weights_.size() != 4) return AsStatus::ALLSPARK_PARAM_ERROR;
    if (in_names_.size() != 2 || out_names_.empty()) return AsStatus::ALLSPARK_PARAM_ERROR;
    AS_CHECK_STATUS(w_.Pack(op_name_, read_wbits(op_proto), read_group(op_proto), weights_[0], weights_[1], weights_[2], stream_of(&ctx)));
    has_norm_ = weights_.size() == 4 && out_names_.size() > 1;
    if (has_norm_) {
      const char* e = attr_ptr(op_proto, "eps");
      if (!e) return AsStatus::ALLSPARK_PARAM_ERROR;
      eps_ = *(const float*)e;
      if ((int)weights_[3]->GetShape()[0] != w_.n || weights_[3]->GetDataType() != w_.ft) return AsStatus::ALLSPARK_PARAM_ERROR;
      tensor_map_->at(out_names_[1])->SetDataType(w_.ft);
      rowsq_ = zeroed(op_name_ + ".rowsq", dihip_rowsq_bytes(), stream_of(&ctx));
      if (!rowsq_) return AsStatus::ALLSPARK_MEMORY_ERROR;
    }
    sync_ = zeroed(op_name_ + ".sync", dihip_gemm_lowp_sync_bytes(), stream_of(&ctx));
    if (!sync_) return AsStatus::ALLSPARK_MEMORY_ERROR;
    hip_ctx(&ctx).AdvertiseLayoutPref(in_names_[0], w_.pref(0));
    tensor_map_->at(out_names_[0])->SetDataType(FLOAT32);
    // the attention operator and the qkv projection in front of it (registered by their Init: the list is built in order)
    blk_attn_ = static_cast<AttnBlockAttnPart*>(hip_ctx(&ctx).Producer(in_names_[0] + "\x01attn"));  // (keys carry the interface)
    if (blk_attn_) {
      blk_qkv_ = static_cast<AttnBlockQkvPart*>(hip_ctx(&ctx).Producer(blk_attn_->BlockAttn().qkv_name + "\x01qkv"));
      if (!blk_qkv_) blk_attn_ = nullptr;
    }
    return AsStatus::ALLSPARK_SUCCESS;
  }
  // ONE launch for qkv projection + attention + this projection (see AttnBlockQkvPart): decoder phase, one request, 16-bit cache,
  // int4 weights with a group per k-tile -- what dihip_decode_attn_block_supported says for this GPU
  bool BlockEligible(RuntimeContext* rt) const {
    static const bool enabled = env_on("DIHIP_DECODER_ATTN_BLOCK", true);
    if (!enabled || hip_ctx(ctx_).AttnBlockDisabled() || !blk_attn_ || !blk_qkv_ || !rt || rt->is_context || m_ != 1 || norm_now_) return false;
    const AttnBlockQkv q = blk_qkv_->BlockQkv();
    const AttnBlockAttn a = blk_attn_->BlockAttn();
    if (q.m != 1 || q.act != 0 || a.batch != 1 || a.seq != 1 || !q.w || q.w->wbits != w_.wbits || q.w->group != w_.group || q.w->ft != w_.ft ||
        a.dt != w_.ft || q.w->n != (a.n + 2 * a.g) * a.h || w_.k != a.n * a.h || w_.n != q.w->k)
      return false;
    if (hip_ctx(ctx_).ActLayout(in_names_[0]) != DIHIP_ACT_ROWMAJOR) return false;
    return dihip_decode_attn_block_supported(w_.wbits, w_.group, w_.n, a.n, a.g, a.h, hip_ctx(ctx_).PlanLength(), a.kv_mode, DihipDtype(w_.ft), 1) != 0;
  }
  AsStatus Reshape(RuntimeContext* rt) override {
    AsTensor* x = tensor_map_->at(in_names_[0]).get();
    Shape s = x->GetShape();
    if (s.empty() || (int)s.back() != w_.k || x->GetDataType() != w_.ft) return AsStatus::ALLSPARK_PARAM_ERROR;
    m_ = (int)(x->Count() / w_.k);
    s.back() = w_.n;
    AsTensor* y = tensor_map_->at(out_names_[0]).get();
    y->SetDataType(FLOAT32);
    Shape ys = s;
    AS_CHECK_STATUS(y->SetShape(std::move(ys)));
    norm_now_ = has_norm_ && norm_fuse_active(ctx_, rt, m_);
    if (has_norm_) {
      AsTensor* xn = tensor_map_->at(out_names_[1]).get();
      xn->SetDataType(w_.ft);
      const bool frag = norm_now_ && pref_frag(hip_ctx(ctx_).LayoutPref(out_names_[1]), m_);
      xn_layout_ = frag ? DIHIP_ACT_FRAG32 : DIHIP_ACT_ROWMAJOR;
      hip_ctx(ctx_).SetActLayout(out_names_[1], xn_layout_);
      const size_t bytes = frag ? dihip_act_frag_bytes(m_, w_.n) : (size_t)m_ * w_.n * SizeofType(w_.ft);
      AS_CHECK_STATUS(ensure_capacity_zeroed(xn, bytes, std::move(s), stream_of(ctx_)));
      // deferred RMSNorm: when the kernel that will serve the consumer takes row partials, 16-bit cache (DecodeSession.defer_ln1 / defer_ln2)
      const ActLayoutPref* cp = hip_ctx(ctx_).LayoutPref(out_names_[1]);
      defer_now_ = norm_now_ && cp && cp->bf16 && w_.ft == BFLOAT16 && ctx_->GetCacheMode() == AsCacheMode::AsCacheDefault &&
                   dihip_prenorm_rowsq_supported(cp->wbits, m_, cp->n, cp->k, cp->group, cp->dual, DIHIP_BF16, xn_layout_) != 0;
    }
    block_now_ = BlockEligible(rt);
    if (blk_attn_ && blk_qkv_) {
      blk_qkv_->BlockSkip(block_now_);
      blk_attn_->BlockSkip(block_now_);
    }
    if (block_now_) {
      const AttnBlockAttn a = blk_attn_->BlockAttn();
      hipStream_t s = stream_of(ctx_);
      // shared by the layers (their launches are ordered on one stream): hand-off state, zeroed once
      const size_t sb = dihip_decode_attn_block_sync_bytes(a.n, a.g, a.h);
      auto it = tensor_map_->find("dihip.attn_block_sync");
      if (it == tensor_map_->end() || it->second->GetSizeInByte() < sb) {
        auto t = std::make_shared<AsTensor>("dihip.attn_block_sync", DeviceType::HIP, INT8, Shape{(int64_t)sb});
        if (!t->GetDataPtr() || hipMemsetAsync(t->GetDataPtr(), 0, sb, s) != hipSuccess) return AsStatus::ALLSPARK_MEMORY_ERROR;
        (*tensor_map_)["dihip.attn_block_sync"] = t;
      }
      const size_t wb = dihip_decode_attn_block_workspace_bytes(a.n, a.g, a.h, ctx_->GetModelMaxLength());
      if (a.ws->GetSizeInByte() < wb) AS_CHECK_STATUS(a.ws->SetShape(Shape{(int64_t)wb}));
      // the split plan of this step's launches (HIPContext::PlanLength) may differ from the one the records of the sync buffer were last
      // used with: the record region is cleared HERE, outside the captured step (dihip_decode_attn_block_prepare)
      AsTensor* sy = tensor_map_->at("dihip.attn_block_sync").get();
      AS_CHECK_STATUS(FromDihip(dihip_decode_attn_block_prepare(s, sy->GetDataPtr(), sy->GetSizeInByte(), a.n, a.g, a.h, hip_ctx(ctx_).PlanLength())));
    }
    return grow_workspace(tensor_map_, dihip_gemm_lowp_workspace_bytes(w_.wbits, std::max(m_, 1), w_.n, w_.k, w_.group));
  }
  AsStatus Forward(RuntimeContext*) override {
    AsTensor* x = tensor_map_->at(in_names_[0]).get();
    AsTensor* y = tensor_map_->at(out_names_[0]).get();
    AsTensor* wsp = tensor_map_->at("workspace").get();
    const bool add_residual = ctx_->GetNranks() <= 1 || ctx_->GetRank() == 0;
    const float* h_res = add_residual ? (const float*)tensor_map_->at(in_names_[1])->GetDataPtr() : nullptr;
    const int x_layout = hip_ctx(ctx_).ActLayout(in_names_[0]);
    hipStream_t s = stream_of(ctx_);
    if (block_now_) {
      const AttnBlockQkv q = blk_qkv_->BlockQkv();
      const AttnBlockAttn a = blk_attn_->BlockAttn();
      if (!a.old_lens) return AsStatus::ALLSPARK_INVALID_CALL_ERROR;  // the attention operator's Forward of this step stages them
      AsTensor* sy = tensor_map_->at("dihip.attn_block_sync").get();
      return FromDihip(dihip_decode_attn_block(s, w_.wbits, (const float*)q.h->GetDataPtr(), h_res, (float*)y->GetDataPtr(), q.gamma, q.eps,
                                               q.w->w->GetDataPtr(), q.w->sz->GetDataPtr(), q.bias, w_.w->GetDataPtr(), w_.sz->GetDataPtr(), a.kd, a.vd,
                                               a.old_lens, a.rope_tab, w_.n, a.n, a.g, a.h, w_.group, a.span, a.max_spans, hip_ctx(ctx_).PlanLength(),
                                               a.kv_mode, DihipDtype(w_.ft), a.alpha, a.ws->GetDataPtr(), a.ws->GetSizeInByte(), sy->GetDataPtr(),
                                               sy->GetSizeInByte()));
    }
    if (norm_now_ && defer_now_) {
      AsTensor* xn = tensor_map_->at(out_names_[1]).get();
      int parts = 0;
      const int st = dihip_fused_gemm_addto_prenorm(s, w_.wbits, x->GetDataPtr(), w_.w->GetDataPtr(), w_.sz->GetDataPtr(), h_res, (float*)y->GetDataPtr(),
                                                    m_, w_.n, w_.k, w_.group, wsp->GetDataPtr(), wsp->GetSizeInByte(), sync_->GetDataPtr(),
                                                    DihipDtype(w_.ft), x_layout, weights_[3]->GetDataPtr(), eps_, xn->GetDataPtr(), xn_layout_,
                                                    (float*)rowsq_->GetDataPtr(), rowsq_->GetSizeInByte(), &parts);
      hip_ctx(ctx_).SetRowNorm(out_names_[1], RowNormState{(const float*)rowsq_->GetDataPtr(), parts, eps_});
      return FromDihip(st);
    }
    if (norm_now_) {
      AsTensor* xn = tensor_map_->at(out_names_[1]).get();
      hip_ctx(ctx_).SetRowNorm(out_names_[1], RowNormState{});
      return FromDihip(dihip_fused_gemm_addto_norm(s, w_.wbits, x->GetDataPtr(), w_.w->GetDataPtr(), w_.sz->GetDataPtr(), h_res,
                                                   (float*)y->GetDataPtr(), m_, w_.n, w_.k, w_.group, wsp->GetDataPtr(), wsp->GetSizeInByte(),
                                                   sync_->GetDataPtr(), DihipDtype(w_.ft), x_layout, weights_[3]->GetDataPtr(), eps_,
                                                   xn->GetDataPtr(), xn_layout_));
    }
    return FromDihip(dihip_fused_gemm_addto_ex(s, w_.wbits, x->GetDataPtr(), w_.w->GetDataPtr(), w_.sz->GetDataPtr(), h_res, (float*)y->GetDataPtr(),
                                               m_, w_.n, w_.k, w_.group, wsp->GetDataPtr(), wsp->GetSizeInByte(), sync_->GetDataPtr(),
                                               DihipDtype(w_.ft), x_layout));
  }

 private:
  PackedLowp w_;
  float eps_ = 1e-6f;
  bool has_norm_ = false, norm_now_ = false, block_now_ = false, defer_now_ = false;
  int m_ = 0, xn_layout_ = DIHIP_ACT_ROWMAJOR;
  std::unique_ptr<AsTensor> rowsq_;
  AttnBlockAttnPart* blk_attn_ = nullptr;
  AttnBlockQkvPart* blk_qkv_ = nullptr;
  std::unique_ptr<AsTensor> sync_;
};
REGISTER_OP(DihipGemmAddTo, HIP, DihipGemmAddToOp)

// ================================================================================================== DihipRopeSpanAttn
// Rotary + DecOptMQA|DecOptMHA.  Input: the fused qkv rows BEFORE Rotary.  attrs of both source operators (num_heads,
// multi_query_group_num, rotary_base; alpha).
//   context phase  dihip_rope_qk on the rows in place (position step + t, rotary_op.cpp:301-338), then SpanAttnOpHIP::runContext
//   decoder phase  16-bit cache: ONE launch (dihip_span_attn_decode_fused_sync: Rotary + DecoderCacheAppend + attention + the
//                  split merge); int8 / uint4 cache: the quantising append launch + the decode kernels on lengths + 1
// Device-resident step state (graph replay): the span tables live on the device and are re-uploaded by Alloc only when a request
// claimed a new span or the batch changed; the lengths are read from "dihip.old_seq_lens" / "dihip.new_seq_lens" when a model
// runner keeps them on the device (HIPContext::LensOnDevice), else uploaded per step like SpanAttnOpHIP does.
class DihipRopeSpanAttnOp : public SpanAttnOpHIP, public AttnBlockAttnPart {
 public:
  explicit DihipRopeSpanAttnOp(const std::string& t = "") : SpanAttnOpHIP(t) {}
  ~DihipRopeSpanAttnOp() override {
    if (pos_host_) (void)hipHostFree(pos_host_);
    if (staged_) (void)hipEventDestroy(staged_);
  }

  AsStatus Init(const OperatorProto& op_proto, const DeviceContext& ctx, const TensorMap& weights_map, TensorMap* tensor_map) override {
    AS_CHECK_STATUS(SpanAttnOpHIP::Init(op_proto, ctx, weights_map, tensor_map));
    const char* p = attr_ptr(op_proto, "rotary_base");
    if (p) base_ = *(const float*)p;
    for (const char* k : {"rotary_type", "rotary_pct", "invfreq_type", "ntk_model_embed", "logn_model_embedding", "mrope_section_size",
                          "seqlen_extrapolation", "rope_ratio", "original_max_position_embeddings", "use_weight"})
      if ((p = attr_ptr(op_proto, k))) {  // only the base rotary of the Qwen2 graph; the fusion pass leaves other variants unfused
        const bool neutral = ((std::string(k) == "rotary_pct" || std::string(k) == "seqlen_extrapolation") && *(const float*)p == 1.0f) ||
                             ((std::string(k) == "rotary_type" || std::string(k) == "invfreq_type") && *(const int*)p == 0);
        if (!neutral) return AsStatus::ALLSPARK_PARAM_ERROR;
      }
    if (h_ != 128 || (dtype_ != BFLOAT16 && dtype_ != FLOAT16)) return AsStatus::ALLSPARK_PARAM_ERROR;
    // inv_freq in double, rounded once (decoder.DecodeSession / oracle/glue.py); the unfused Rotary operator keeps the
    // reference's float pow (rotary_op.h:51-76) -- at most one ulp apart
    std::vector<float> inv(h_ / 2);
    for (int i = 0; i < h_ / 2; ++i) inv[i] = (float)(1.0 / std::pow((double)base_, (double)(2 * i) / (double)h_));
    AsTensor* shared = SharedTensor("dihip.inv_freq", FLOAT32, Shape{(int64_t)inv.size()});
    if (!shared || !shared->GetDataPtr()) return AsStatus::ALLSPARK_MEMORY_ERROR;
    if (hipMemcpy(shared->GetDataPtr(), inv.data(), inv.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
      return AsStatus::ALLSPARK_RUNTIME_ERROR;
    if (hipEventCreateWithFlags(&staged_, hipEventDisableTiming) != hipSuccess) return AsStatus::ALLSPARK_RUNTIME_ERROR;
    hip_ctx(&ctx).RegisterProducer(out_names_[0] + "\x01attn", static_cast<AttnBlockAttnPart*>(this));
    return AsStatus::ALLSPARK_SUCCESS;
  }
  AttnBlockAttn BlockAttn() const override {
    AttnBlockAttn a;
    a.kd = k_arr_dev_ ? reinterpret_cast<void* const*>(k_arr_dev_->GetDataPtr()) : nullptr;
    a.vd = v_arr_dev_ ? reinterpret_cast<void* const*>(v_arr_dev_->GetDataPtr()) : nullptr;
    a.old_lens = blk_old_lens_;
    auto tab = tensor_map_->find("dihip.rope_table");
    a.rope_tab = tab == tensor_map_->end() ? nullptr : (const float*)tab->second->GetDataPtr();
    auto ws = tensor_map_->find("dihip.attn_ws");
    a.ws = ws == tensor_map_->end() ? nullptr : ws->second.get();
    a.n = n_;
    a.g = g_;
    a.h = h_;
    a.span = span_;
    a.max_spans = max_spans_;
    a.kv_mode = kv_mode_;
    a.batch = batch_;
    a.seq = seq_;
    a.dt = dtype_;
    a.alpha = alpha_;
    a.qkv_name = in_names_[0];
    return a;
  }
  void BlockSkip(bool skip) override { skip_ = skip; }

  AsStatus Reshape(RuntimeContext* rt) override {
    skip_ = false;  // (the o-projection's Reshape, later in the list, decides anew)
    blk_old_lens_ = nullptr;
    AS_CHECK_STATUS(SpanAttnOpHIP::Reshape(rt));
    const int nb = std::max(batch_, 1), max_len = ctx_->GetModelMaxLength();
    // shared between the layers: rope table (cos, sin per position), attention workspace, arrival tickets
    AsTensor* tab = SharedTensor("dihip.rope_table", FLOAT32, Shape{(int64_t)(max_len + 1), (int64_t)h_ / 2, 2});
    if (!tab || !tab->GetDataPtr()) return AsStatus::ALLSPARK_MEMORY_ERROR;
    if (!hip_ctx(ctx_).ActLayout("dihip.rope_table.ready")) {
      AS_CHECK_STATUS(FromDihip(dihip_rope_table(Stream(), (float*)tab->GetDataPtr(), (const float*)tensor_map_->at("dihip.inv_freq")->GetDataPtr(),
                                                 max_len + 1, h_)));
      hip_ctx(ctx_).SetActLayout("dihip.rope_table.ready", 1);
    }
    const size_t ws = std::max<size_t>({dihip_span_attn_decode_workspace_bytes(nb, n_, h_, max_len, 0),
                                        dihip_span_attn_fused_workspace_bytes(nb, n_, g_, h_, max_len), (size_t)256});
    AsTensor* aws = SharedTensor("dihip.attn_ws", INT8, Shape{(int64_t)ws});
    if (!aws) return AsStatus::ALLSPARK_MEMORY_ERROR;
    if (aws->GetSizeInByte() < ws) AS_CHECK_STATUS(aws->SetShape(Shape{(int64_t)ws}));
    const size_t sb = dihip_span_attn_sync_bytes(nb, n_);
    AsTensor* sy = SharedTensor("dihip.attn_sync", INT8, Shape{(int64_t)sb});
    if (!sy) return AsStatus::ALLSPARK_MEMORY_ERROR;
    if (sy->GetSizeInByte() < sb || !hip_ctx(ctx_).ActLayout("dihip.attn_sync.zeroed")) {
      AS_CHECK_STATUS(sy->SetShape(Shape{(int64_t)std::max(sb, sy->GetSizeInByte())}));
      if (hipMemsetAsync(sy->GetDataPtr(), 0, sy->GetSizeInByte(), Stream()) != hipSuccess) return AsStatus::ALLSPARK_RUNTIME_ERROR;
      hip_ctx(ctx_).SetActLayout("dihip.attn_sync.zeroed", 1);
    }
    // the output may go out in FRAG32 for the o projection (quantised caches; DecodeSession.attn_frag)
    AsTensor* out = tensor_map_->at(out_names_[0]).get();
    const bool frag = !rt->is_context && kv_mode_ != 0 && batch_ <= 32 && pref_frag(hip_ctx(ctx_).LayoutPref(out_names_[0]), batch_);
    out_layout_ = frag ? DIHIP_ACT_FRAG32 : DIHIP_ACT_ROWMAJOR;
    hip_ctx(ctx_).SetActLayout(out_names_[0], out_layout_);
    if (frag) AS_CHECK_STATUS(ensure_capacity_zeroed(out, dihip_act_frag_bytes(batch_, n_ * h_), Shape{batch_, seq_, (int64_t)n_ * h_}, Stream()));
    if (rt->is_context) {
      if ((size_t)seq_ > pos_cap_) {
        if (pos_host_) (void)hipHostFree(pos_host_);
        pos_cap_ = std::max<size_t>(seq_, 256);
        if (hipHostMalloc((void**)&pos_host_, pos_cap_ * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) return AsStatus::ALLSPARK_MEMORY_ERROR;
      }
      if (!pos_dev_) pos_dev_ = std::make_unique<AsTensor>(op_name_ + ".positions", DeviceType::HIP, INT32, Shape{(int64_t)seq_});
      AS_CHECK_STATUS(pos_dev_->SetShape(Shape{(int64_t)std::max(seq_, 1)}));
    }
    tables_valid_ = false;  // batch membership may have changed
    return AsStatus::ALLSPARK_SUCCESS;
  }

  // SpanAttnOp::Alloc + (decoder phase) the span tables of the batch on the device, uploaded only when they changed
  AsStatus Alloc(RuntimeContext* rt) override {
    AS_CHECK_STATUS(SpanAttnOpHIP::Alloc(rt));
    if (rt->is_context) return AsStatus::ALLSPARK_SUCCESS;
    bool changed = !tables_valid_;
    if ((int)staged_spans_.size() != batch_) staged_spans_.assign(batch_, 0), changed = true;
    for (int b = 0; b < batch_ && !changed; ++b)
      if ((rt->GetGenCtx(b)->step + 1 + span_ - 1) / span_ != staged_spans_[b]) changed = true;
    if (!changed) return AsStatus::ALLSPARK_SUCCESS;
    if (hipEventSynchronize(staged_) != hipSuccess) return AsStatus::ALLSPARK_RUNTIME_ERROR;  // the previous upload left the staging rows
    for (int b = 0; b < batch_; ++b) {
      const GenerateContext* gc = rt->GetGenCtx(b);
      AS_CHECK_STATUS(stageSpans(gc, b, gc->step + 1));
      staged_spans_[b] = (gc->step + 1 + span_ - 1) / span_;
    }
    const size_t bytes = (size_t)batch_ * max_spans_ * sizeof(void*);
    if (hipMemcpyAsync(k_arr_dev_->GetDataPtr(), k_arr_host_->GetDataPtr(), bytes, hipMemcpyHostToDevice, Stream()) != hipSuccess ||
        hipMemcpyAsync(v_arr_dev_->GetDataPtr(), v_arr_host_->GetDataPtr(), bytes, hipMemcpyHostToDevice, Stream()) != hipSuccess ||
        hipEventRecord(staged_, Stream()) != hipSuccess)
      return AsStatus::ALLSPARK_RUNTIME_ERROR;
    tables_valid_ = true;
    return AsStatus::ALLSPARK_SUCCESS;
  }

  AsStatus Forward(RuntimeContext* rt) override {
    if (rt->is_context) return RopeThenContext(rt);
    if (rt->GetGenCtxListSize() != batch_ || seq_ != 1) return AsStatus::ALLSPARK_PARAM_ERROR;
    const uint32_t *old_lens, *new_lens;
    if (hip_ctx(ctx_).LensOnDevice()) {
      auto o = tensor_map_->find("dihip.old_seq_lens"), n = tensor_map_->find("dihip.new_seq_lens");
      if (o == tensor_map_->end() || n == tensor_map_->end()) return AsStatus::ALLSPARK_INVALID_CALL_ERROR;
      old_lens = (const uint32_t*)o->second->GetDataPtr();
      new_lens = (const uint32_t*)n->second->GetDataPtr();
    } else {
      int32_t* lens = reinterpret_cast<int32_t*>(lens_host_->GetDataPtr());
      for (int b = 0; b < batch_; ++b) {
        lens[b] = rt->GetGenCtx(b)->step;
        lens[batch_ + b] = rt->GetGenCtx(b)->step + 1;
      }
      if (hipMemcpyAsync(lens_dev_->GetDataPtr(), lens, (size_t)2 * batch_ * sizeof(int32_t), hipMemcpyHostToDevice, Stream()) != hipSuccess)
        return AsStatus::ALLSPARK_RUNTIME_ERROR;
      old_lens = (const uint32_t*)lens_dev_->GetDataPtr();
      new_lens = old_lens + batch_;
    }
    blk_old_lens_ = old_lens;
    if (skip_) return AsStatus::ALLSPARK_SUCCESS;  // part of the o-projection's single launch (AttnBlockAttnPart): lengths staged, nothing launched
    void* const* kd = reinterpret_cast<void* const*>(k_arr_dev_->GetDataPtr());
    void* const* vd = reinterpret_cast<void* const*>(v_arr_dev_->GetDataPtr());
    const void* qkv = tensor_map_->at(in_names_[0])->GetDataPtr();
    void* out = tensor_map_->at(out_names_[0])->GetDataPtr();
    AsTensor* aws = tensor_map_->at("dihip.attn_ws").get();
    AsTensor* sy = tensor_map_->at("dihip.attn_sync").get();
    static const bool merge_in_launch = [] {
      const char* e = getenv("DIHIP_DECODER_ATTN_MERGE");
      return !(e && std::string(e) == "launch");
    }();
    const int max_len = hip_ctx(ctx_).PlanLength();  // (the launch plan's length: the runner's bucket, <= the engine's maximum)
    if (kv_mode_ == 0) {
      return FromDihip(dihip_span_attn_decode_fused_sync(Stream(), out, qkv, kd, vd, old_lens, (const float*)tensor_map_->at("dihip.rope_table")->GetDataPtr(),
                                                         batch_, n_, g_, h_, span_, max_spans_, max_len, kv_mode_, DihipDtype(dtype_), alpha_,
                                                         aws->GetDataPtr(), aws->GetSizeInByte(), merge_in_launch ? sy->GetDataPtr() : nullptr,
                                                         merge_in_launch ? sy->GetSizeInByte() : 0));
    }
    static const bool u4_step = env_on("DIHIP_ATTN_U4_FUSED", true);  // decoder.DecodeSession.step_attention
    static const bool i8_step = env_on("DIHIP_ATTN_I8_FUSED", true);
    if ((kv_mode_ == DIHIP_KV_U4 && dtype_ == BFLOAT16 && u4_step) || (kv_mode_ == DIHIP_KV_I8 && i8_step && batch_ <= 4)) {  // (int8: where it pays, as DecodeSession)
      // uint4 cache with bf16 rows, int8 cache: one launch as well (Rotary + quantising append + attention + merge), FRAG32 output included
      return FromDihip(dihip_span_attn_decode_step(Stream(), out, qkv, kd, vd, old_lens, (const float*)tensor_map_->at("dihip.rope_table")->GetDataPtr(),
                                                   batch_, n_, g_, h_, span_, max_spans_, max_len, kv_mode_, DihipDtype(dtype_), alpha_,
                                                   aws->GetDataPtr(), aws->GetSizeInByte(), merge_in_launch ? sy->GetDataPtr() : nullptr,
                                                   merge_in_launch ? sy->GetSizeInByte() : 0, out_layout_));
    }
    AS_CHECK_STATUS(FromDihip(dihip_rope_kv_append(Stream(), kd, vd, q_dev_->GetDataPtr(), qkv, old_lens,
                                                   (const float*)tensor_map_->at("dihip.inv_freq")->GetDataPtr(), batch_, n_, g_, h_, span_,
                                                   max_spans_, kv_mode_, DihipDtype(dtype_))));
    return FromDihip(dihip_span_attn_decode_sync(Stream(), out, q_dev_->GetDataPtr(), (const void* const*)kd, (const void* const*)vd, new_lens,
                                                 batch_, n_, g_, h_, span_, max_spans_, max_len, kv_mode_, DihipDtype(dtype_), alpha_,
                                                 aws->GetDataPtr(), aws->GetSizeInByte(), sy->GetDataPtr(), sy->GetSizeInByte(), out_layout_));
  }

 private:
  AsTensor* SharedTensor(const char* name, DataType dt, const Shape& shape) {
    auto it = tensor_map_->find(name);
    if (it == tensor_map_->end()) it = tensor_map_->emplace(name, std::make_shared<AsTensor>(name, DeviceType::HIP, dt, shape)).first;
    return it->second.get();
  }
  AsStatus RopeThenContext(RuntimeContext* rt) {
    const GenerateContext* gc = rt->GetContextGenCtx();
    if (hipEventSynchronize(staged_) != hipSuccess) return AsStatus::ALLSPARK_RUNTIME_ERROR;
    for (int t = 0; t < seq_; ++t) pos_host_[t] = (uint32_t)(gc->step + t);  // rotary_op.cpp:331: step, which is prefix_len here
    if (hipMemcpyAsync(pos_dev_->GetDataPtr(), pos_host_, (size_t)seq_ * sizeof(uint32_t), hipMemcpyHostToDevice, Stream()) != hipSuccess ||
        hipEventRecord(staged_, Stream()) != hipSuccess)
      return AsStatus::ALLSPARK_RUNTIME_ERROR;
    AS_CHECK_STATUS(FromDihip(dihip_rope_qk(Stream(), tensor_map_->at(in_names_[0])->GetDataPtr(), (const uint32_t*)pos_dev_->GetDataPtr(),
                                            (const float*)tensor_map_->at("dihip.inv_freq")->GetDataPtr(), seq_, n_, g_, h_, DihipDtype(dtype_))));
    tables_valid_ = false;  // runContext stages this request's spans into row 0 of the tables
    return runContext(rt);
  }

  float base_ = 10000.f;
  int out_layout_ = DIHIP_ACT_ROWMAJOR;
  bool skip_ = false;
  const uint32_t* blk_old_lens_ = nullptr;
  bool tables_valid_ = false;
  std::vector<int> staged_spans_;
  hipEvent_t staged_ = nullptr;
  uint32_t* pos_host_ = nullptr;
  size_t pos_cap_ = 0;
  std::unique_ptr<AsTensor> pos_dev_;
};
REGISTER_OP(DihipRopeSpanAttn, HIP, DihipRopeSpanAttnOp)

// ====================================================================================================== DihipMoeBlock
// The feed-forward block of a mixture-of-experts layer (python/pyhie/allspark/model/qwen_v20_moe.py:318-382), twelve operators
// in the reference graph:
//   LayerNormNoBeta -> Gemm "mlp.gate" (router) -> MOE -> [AllReduce] -> GemmA16Wx "shared_expert.gate_up_proj" -> UnaryGLU ->
//   GemmA16Wx "shared_expert.down_proj" ; Gemm "shared_expert_gate" (SIGMOID) -> CalcExpert -> [AllReduce] -> Binary ADD -> Binary ADD
// as the launches of decoder.DecodeSession._moe_block: norm -> router + shared gate in one launch -> routing (+ slot grouping) ->
// expert GEMVs -> shared expert SwiGLU pair -> shared down projection -> combine (finalize-routing + CalcExpert + both adds) on
// the f32 hidden rows.  Under expert / tensor parallelism the rank adds its partial sums (and, on rank 0, the residual) first;
// the fusion pass leaves ONE AllReduce of the f32 rows behind the operator -- the same sum as the reference's two.
//   inputs  [h f32 [.., hidden]]      outputs [h_out f32]
//   weights [ffn gamma, mlp.gate W [hidden, E], experts gate_up W/scales/zeros, experts down W/scales/zeros (the MOEA16W8 stacks),
//            shared gate_up W/scales/zeros ([hidden, 2 I], columns [gate | up]), shared down W/scales/zeros, shared_expert_gate W [hidden, 1]]
//   attrs   eps, num_experts, num_experts_per_tok, [use_ep], [GroupSize] (experts), wbits / [shared.GroupSize] (shared expert)
class DihipMoeBlockOp : public MoeA16W8HIP {
 public:
  explicit DihipMoeBlockOp(const std::string& t = "") : MoeA16W8HIP(t) {}
  AsStatus InitV2(const OperatorProto& op_proto, const DeviceContext& ctx, const TensorMap& weights_map, TensorMap& weights_buffer,
                  TensorMap* tensor_map, RuntimeContext* runtime_ctx) override {
    (void)weights_buffer;
    (void)runtime_ctx;
    AS_CHECK_STATUS(AsOperator::Init(op_proto, ctx, weights_map, tensor_map));
    if (ctx.GetDeviceType() != DeviceType::HIP || weights_.size() != 15 || in_names_.size() != 1) return AsStatus::ALLSPARK_PARAM_ERROR;
    const char* e = attr_ptr(op_proto, "eps");
    if (!e) return AsStatus::ALLSPARK_PARAM_ERROR;
    eps_ = *(const float*)e;
    AS_CHECK_STATUS(ParseMoeAttrs(op_proto, ctx));
    AS_CHECK_STATUS(PackExperts(weights_.data() + 2, ctx));
    hipStream_t s = stream_of(&ctx);
    // the two unquantised skinny weights
    const AsTensor *router = weights_[1], *sig = weights_[14];
    if (router->GetShape().size() != 2 || (int)router->GetShape()[0] != hidden_ || (int)router->GetShape()[1] != num_expert_ ||
        router->GetDataType() != ftype_)
      return AsStatus::ALLSPARK_PARAM_ERROR;
    if (sig->Count() != hidden_ || sig->GetDataType() != ftype_) return AsStatus::ALLSPARK_PARAM_ERROR;
    if ((int)weights_[0]->GetShape()[0] != hidden_ || weights_[0]->GetDataType() != ftype_) return AsStatus::ALLSPARK_PARAM_ERROR;
    router_ = std::make_unique<AsTensor>(op_name_ + ".router_packed", DeviceType::HIP, INT8, Shape{(int64_t)dihip_dense_packed_weight_bytes(num_expert_, hidden_)});
    sig_ = std::make_unique<AsTensor>(op_name_ + ".shared_gate_packed", DeviceType::HIP, INT8, Shape{(int64_t)dihip_dense_packed_weight_bytes(1, hidden_)});
    if (!router_->GetDataPtr() || !sig_->GetDataPtr()) return AsStatus::ALLSPARK_MEMORY_ERROR;
    AS_CHECK_STATUS(FromDihip(dihip_dense_pack(s, router->GetDataPtr(), num_expert_, hidden_, DihipDtype(ftype_), router_->GetDataPtr())));
    AS_CHECK_STATUS(FromDihip(dihip_dense_pack(s, sig->GetDataPtr(), 1, hidden_, DihipDtype(ftype_), sig_->GetDataPtr())));
    // the shared expert: column halves of gate_up_proj become the gate and the up tensor (UnaryGLU, unary.cu:122-132)
    const int wb = read_wbits(op_proto);
    const char* sg = attr_ptr(op_proto, "shared.GroupSize");
    const int grp = sg ? *(const int*)sg : -1;
    const AsTensor *gu = weights_[8], *gus = weights_[9], *guz = weights_[10];
    if (gu->GetShape().size() != 2 || gus->GetShape().size() != 2 || (int)gu->GetShape()[0] != hidden_ || gus->GetDataType() != ftype_ ||
        (gus->GetShape()[1] % 4))
      return AsStatus::ALLSPARK_PARAM_ERROR;
    const int inter = (int)gus->GetShape()[1] / 2, G = (int)gus->GetShape()[0];
    const size_t wrow = wb == 4 ? (size_t)inter / 2 : (size_t)inter;  // bytes of one half of a weight row
    AsTensor tw(op_name_ + ".tmp_w", DeviceType::HIP, gu->GetDataType(), Shape{hidden_, (int64_t)wrow});
    AsTensor ts(op_name_ + ".tmp_s", DeviceType::HIP, ftype_, Shape{G, inter}), tz(op_name_ + ".tmp_z", DeviceType::HIP, ftype_, Shape{G, inter});
    if (!tw.GetDataPtr() || !ts.GetDataPtr() || !tz.GetDataPtr()) return AsStatus::ALLSPARK_MEMORY_ERROR;
    for (int half = 0; half < 2; ++half) {
      if (hipMemcpy2DAsync(tw.GetDataPtr(), wrow, (const char*)gu->GetDataPtr() + half * wrow, 2 * wrow, wrow, hidden_, hipMemcpyDeviceToDevice, s) != hipSuccess ||
          hipMemcpy2DAsync(ts.GetDataPtr(), (size_t)inter * 2, (const char*)gus->GetDataPtr() + (size_t)half * inter * 2, (size_t)inter * 4, (size_t)inter * 2, G,
                           hipMemcpyDeviceToDevice, s) != hipSuccess ||
          hipMemcpy2DAsync(tz.GetDataPtr(), (size_t)inter * 2, (const char*)guz->GetDataPtr() + (size_t)half * inter * 2, (size_t)inter * 4, (size_t)inter * 2, G,
                           hipMemcpyDeviceToDevice, s) != hipSuccess)
        return AsStatus::ALLSPARK_RUNTIME_ERROR;
      AS_CHECK_STATUS((half ? su_ : sg_).Pack(op_name_ + (half ? ".shared_up" : ".shared_gate"), wb, grp, &tw, &ts, &tz, s, false));
    }
    AS_CHECK_STATUS(sd_.Pack(op_name_ + ".shared_down", wb, grp, weights_[11], weights_[12], weights_[13], s));
    if (sg_.n != inter || sg_.k != hidden_ || sd_.k != inter || sd_.n != hidden_ || sd_.ft != ftype_) return AsStatus::ALLSPARK_PARAM_ERROR;
    sync_ = zeroed(op_name_ + ".sync", dihip_gemm_lowp_sync_bytes(), s);
    if (!sync_) return AsStatus::ALLSPARK_MEMORY_ERROR;
    if (hipStreamSynchronize(s) != hipSuccess) return AsStatus::ALLSPARK_RUNTIME_ERROR;  // the staging tensors die here
    tensor_map_->at(out_names_[0])->SetDataType(FLOAT32);
    return AsStatus::ALLSPARK_SUCCESS;
  }

  AsStatus Reshape(RuntimeContext* rt) override {
    AsTensor* h = tensor_map_->at(in_names_[0]).get();
    Shape s = h->GetShape();
    if (s.empty() || (int)s.back() != hidden_ || h->GetDataType() != FLOAT32) return AsStatus::ALLSPARK_PARAM_ERROR;
    total_token_ = (int)(h->Count() / hidden_);
    const int t = std::max(total_token_, 1);
    AsTensor* y = tensor_map_->at(out_names_[0]).get();
    y->SetDataType(FLOAT32);
    AS_CHECK_STATUS(y->SetShape(std::move(s)));
    static const bool fuse_on = env_on("DIHIP_MOE_FUSED", true), frag_on = env_on("DIHIP_MOE_ACT_FRAG", true);
    grouped_ = fuse_on && total_token_ > 1 && (int64_t)total_token_ * top_k_ <= 2048;  // decoder.DecodeSession._moe_block
    act_frag_ = frag_on && rt && !rt->is_context && total_token_ > 4 && total_token_ <= 32 && sg_.n % 32 == 0;
    // scratch rows shared by the layers' blocks (one block runs at a time)
    struct Need { const char* name; DataType dt; int64_t count; AsTensor** slot; };
    const size_t act_bytes = act_frag_ ? dihip_act_frag_bytes(total_token_, sg_.n) : (size_t)t * sg_.n * 2;
    const Need needs[] = {{"dihip.moe_xn", ftype_, (int64_t)t * hidden_, &xn_},      {"dihip.moe_logits", ftype_, (int64_t)t * num_expert_, &logits_},
                          {"dihip.moe_sig", ftype_, (int64_t)t, &sigv_},              {"dihip.moe_scores", FLOAT32, (int64_t)t * top_k_, &scores_},
                          {"dihip.moe_experts", INT32, (int64_t)t * top_k_, &idx_},   {"dihip.moe_out", ftype_, (int64_t)t * hidden_, &moe_out_},
                          {"dihip.moe_shared", ftype_, (int64_t)t * hidden_, &shared_}, {"dihip.moe_ws", INT8, (int64_t)dihip_moe_workspace_bytes(t, top_k_, hidden_, proj_), &mws_}};
    for (const Need& n : needs) {
      auto it = tensor_map_->find(n.name);
      if (it == tensor_map_->end()) it = tensor_map_->emplace(n.name, std::make_shared<AsTensor>(n.name, DeviceType::HIP, n.dt, Shape{n.count})).first;
      if (it->second->Count() < n.count) AS_CHECK_STATUS(it->second->SetShape(Shape{n.count}));
      if (!it->second->GetDataPtr()) return AsStatus::ALLSPARK_MEMORY_ERROR;
      *n.slot = it->second.get();
    }
    {  // SiLU(gate) * up of the shared expert: FRAG32 for small decode batches (zero padding rows), row-major otherwise
      auto it = tensor_map_->find("dihip.moe_act");
      if (it == tensor_map_->end()) it = tensor_map_->emplace("dihip.moe_act", std::make_shared<AsTensor>("dihip.moe_act", DeviceType::HIP, ftype_, Shape{0})).first;
      act_ = it->second.get();
      if (act_->GetSizeInByte() < act_bytes || hip_ctx(ctx_).ActLayout("dihip.moe_act") != (act_frag_ ? DIHIP_ACT_FRAG32 : DIHIP_ACT_ROWMAJOR)) {
        AS_CHECK_STATUS(act_->SetShape(Shape{(int64_t)((std::max(act_bytes, act_->GetSizeInByte()) + 1) / 2)}));
        if (hipMemsetAsync(act_->GetDataPtr(), 0, act_->GetSizeInByte(), stream_of(ctx_)) != hipSuccess) return AsStatus::ALLSPARK_RUNTIME_ERROR;
        hip_ctx(ctx_).SetActLayout("dihip.moe_act", act_frag_ ? DIHIP_ACT_FRAG32 : DIHIP_ACT_ROWMAJOR);
      }
    }
    const size_t ws = std::max(dihip_gemm_lowp_workspace_bytes(sg_.wbits, t, sg_.n, sg_.k, sg_.group),
                               dihip_gemm_lowp_workspace_bytes(sd_.wbits, t, sd_.n, sd_.k, sd_.group));
    return grow_workspace(tensor_map_, ws);
  }

  AsStatus Forward(RuntimeContext*) override {
    const float* h = (const float*)tensor_map_->at(in_names_[0])->GetDataPtr();
    float* y = (float*)tensor_map_->at(out_names_[0])->GetDataPtr();
    AsTensor* wsp = tensor_map_->at("workspace").get();
    hipStream_t s = stream_of(ctx_);
    const int T = total_token_, dt = DihipDtype(ftype_);
    AS_CHECK_STATUS(FromDihip(dihip_rmsnorm_rows(s, xn_->GetDataPtr(), h, weights_[0]->GetDataPtr(), eps_, T, hidden_, dt)));
    AS_CHECK_STATUS(FromDihip(dihip_moe_router_gate(s, xn_->GetDataPtr(), router_->GetDataPtr(), sig_->GetDataPtr(), logits_->GetDataPtr(),
                                                    sigv_->GetDataPtr(), T, num_expert_, hidden_, dt)));
    float* scores = (float*)scores_->GetDataPtr();
    int32_t* idx = (int32_t*)idx_->GetDataPtr();
    if (grouped_) {
      AS_CHECK_STATUS(FromDihip(dihip_moe_route_grouped(s, logits_->GetDataPtr(), T, num_expert_, top_k_, scores, idx, dt, ep_first_, ep_num_, hidden_,
                                                        proj_, mws_->GetDataPtr(), mws_->GetSizeInByte())));
    } else {
      AS_CHECK_STATUS(FromDihip(dihip_moe_route_ep(s, logits_->GetDataPtr(), T, num_expert_, top_k_, scores, idx, dt, ep_first_, ep_num_)));
    }
    AS_CHECK_STATUS(FromDihip(dihip_moe_experts_ex(s, 8, xn_->GetDataPtr(), idx, scores, gate_w_->GetDataPtr(), gate_sz_->GetDataPtr(), up_w_->GetDataPtr(),
                                                   up_sz_->GetDataPtr(), down_w_->GetDataPtr(), down_sz_->GetDataPtr(), T, top_k_, hidden_, proj_,
                                                   group_size_, moe_out_->GetDataPtr(), mws_->GetDataPtr(), mws_->GetSizeInByte(), dt,
                                                   grouped_ ? (DIHIP_MOE_PREGROUPED | DIHIP_MOE_NO_FINALIZE) : 0)));
    const int lay = act_frag_ ? DIHIP_ACT_FRAG32 : DIHIP_ACT_ROWMAJOR;
    AS_CHECK_STATUS(FromDihip(dihip_prenorm_swiglu(s, sg_.wbits, xn_->GetDataPtr(), DIHIP_ACT_ROWMAJOR, sg_.w->GetDataPtr(), sg_.sz->GetDataPtr(),
                                                   su_.w->GetDataPtr(), su_.sz->GetDataPtr(), act_->GetDataPtr(), T, sg_.n, sg_.k, sg_.group,
                                                   wsp->GetDataPtr(), wsp->GetSizeInByte(), sync_->GetDataPtr(), dt, lay)));
    if (act_frag_) {
      AS_CHECK_STATUS(FromDihip(dihip_prenorm_gemm(s, sd_.wbits, act_->GetDataPtr(), DIHIP_ACT_FRAG32, sd_.w->GetDataPtr(), sd_.sz->GetDataPtr(), nullptr,
                                                   shared_->GetDataPtr(), T, sd_.n, sd_.k, sd_.group, 0, wsp->GetDataPtr(), wsp->GetSizeInByte(),
                                                   sync_->GetDataPtr(), dt)));
    } else {
      auto gemm = sd_.wbits == 8 ? dihip_gemm_a16w8 : dihip_gemm_a16w4;
      AS_CHECK_STATUS(FromDihip(gemm(s, act_->GetDataPtr(), sd_.w->GetDataPtr(), sd_.sz->GetDataPtr(), nullptr, nullptr, shared_->GetDataPtr(), T, sd_.n,
                                     sd_.k, sd_.group, 0, 1.0f, wsp->GetDataPtr(), wsp->GetSizeInByte(), sync_->GetDataPtr(), dt)));
    }
    const float* h_res = (ctx_->GetNranks() <= 1 || ctx_->GetRank() == 0) ? h : nullptr;  // the residual rides on rank 0 (gemm_op.cpp:133-137)
    if (grouped_)
      return FromDihip(dihip_moe_combine(s, y, h_res, mws_->GetDataPtr(), scores, idx, shared_->GetDataPtr(), sigv_->GetDataPtr(), T, top_k_, hidden_,
                                         proj_, dt));
    return FromDihip(dihip_moe_shared_combine(s, y, h_res, moe_out_->GetDataPtr(), shared_->GetDataPtr(), sigv_->GetDataPtr(), T, hidden_, dt));
  }

 private:
  float eps_ = 1e-6f;
  bool grouped_ = false, act_frag_ = false;
  PackedLowp sg_, su_, sd_;
  std::unique_ptr<AsTensor> router_, sig_, sync_;
  AsTensor *xn_ = nullptr, *logits_ = nullptr, *sigv_ = nullptr, *scores_ = nullptr, *idx_ = nullptr, *moe_out_ = nullptr, *shared_ = nullptr,
           *mws_ = nullptr, *act_ = nullptr;
};
REGISTER_OP(DihipMoeBlock, HIP, DihipMoeBlockOp)

// ===================================================================================================== DihipFinalNorm
// LayerNormNoBeta of the f32 hidden rows into FT rows (dihip_rmsnorm_rows): the seam between the fused layers and a tail that stays
// on the reference's own operators (tensor-parallel lm_head: GetLastLine -> Gemm(splitk) -> AllReduce -> GenerateOp).
class DihipFinalNormOp : public AsOperator {
 public:
  explicit DihipFinalNormOp(const std::string& t = "") : AsOperator(t) {}
  AsStatus Init(const OperatorProto& op_proto, const DeviceContext& ctx, const TensorMap& weights_map, TensorMap* tensor_map) override {
    AS_CHECK_STATUS(AsOperator::Init(op_proto, ctx, weights_map, tensor_map));
    if (weights_.size() != 1) return AsStatus::ALLSPARK_PARAM_ERROR;
    const char* e = attr_ptr(op_proto, "eps");
    if (!e) return AsStatus::ALLSPARK_PARAM_ERROR;
    eps_ = *(const float*)e;
    ft_ = weights_[0]->GetDataType();
    if (ft_ != BFLOAT16 && ft_ != FLOAT16) return AsStatus::ALLSPARK_PARAM_ERROR;
    hidden_ = (int)weights_[0]->GetShape()[0];
    tensor_map_->at(out_names_[0])->SetDataType(ft_);
    return AsStatus::ALLSPARK_SUCCESS;
  }
  AsStatus Reshape(RuntimeContext*) override {
    AsTensor* h = tensor_map_->at(in_names_[0]).get();
    Shape s = h->GetShape();
    if (s.empty() || (int)s.back() != hidden_ || h->GetDataType() != FLOAT32) return AsStatus::ALLSPARK_PARAM_ERROR;
    rows_ = (int)(h->Count() / hidden_);
    AsTensor* y = tensor_map_->at(out_names_[0]).get();
    y->SetDataType(ft_);
    return y->SetShape(std::move(s));
  }
  AsStatus Forward(RuntimeContext*) override {
    return FromDihip(dihip_rmsnorm_rows(stream_of(ctx_), tensor_map_->at(out_names_[0])->GetDataPtr(),
                                        (const float*)tensor_map_->at(in_names_[0])->GetDataPtr(), weights_[0]->GetDataPtr(), eps_, rows_,
                                        hidden_, DihipDtype(ft_)));
  }

 private:
  int hidden_ = 0, rows_ = 0;
  float eps_ = 1e-6f;
  DataType ft_ = BFLOAT16;
};
REGISTER_OP(DihipFinalNorm, HIP, DihipFinalNormOp)

// ======================================================================================================== DihipLMHead
// logits (f32) = RMSNorm(h_last; gamma, eps) . W_lm: final LayerNormNoBeta + GetLastLine + the lm_head Gemm
// (model_base.py:690-703; qwen_v15.py:383-388).  weights [gamma, lm_head.weight FT [hidden, vocab]]; one rank.
class DihipLMHeadOp : public AsOperator {
 public:
  explicit DihipLMHeadOp(const std::string& t = "") : AsOperator(t) {}
  AsStatus Init(const OperatorProto& op_proto, const DeviceContext& ctx, const TensorMap& weights_map, TensorMap* tensor_map) override {
    AS_CHECK_STATUS(AsOperator::Init(op_proto, ctx, weights_map, tensor_map));
    if (ctx.GetDeviceType() != DeviceType::HIP || weights_.size() != 2 || ctx.GetNranks() > 1) return AsStatus::ALLSPARK_PARAM_ERROR;
    const char* e = attr_ptr(op_proto, "eps");
    if (!e) return AsStatus::ALLSPARK_PARAM_ERROR;
    eps_ = *(const float*)e;
    const AsTensor* w = weights_[1];
    if (w->GetShape().size() != 2) return AsStatus::ALLSPARK_PARAM_ERROR;
    k_ = (int)w->GetShape()[0];
    n_ = (int)w->GetShape()[1];
    ft_ = w->GetDataType();
    if ((ft_ != BFLOAT16 && ft_ != FLOAT16) || weights_[0]->GetDataType() != ft_ || (int)weights_[0]->GetShape()[0] != k_)
      return AsStatus::ALLSPARK_PARAM_ERROR;
    packed_ = std::make_unique<AsTensor>(op_name_ + ".packed_w", DeviceType::HIP, INT8, Shape{(int64_t)dihip_dense_packed_weight_bytes(n_, k_)});
    if (!packed_->GetDataPtr()) return AsStatus::ALLSPARK_MEMORY_ERROR;
    sync_ = zeroed(op_name_ + ".sync", dihip_gemm_lowp_sync_bytes(), stream_of(&ctx));
    if (!sync_) return AsStatus::ALLSPARK_MEMORY_ERROR;
    tensor_map_->at(out_names_[0])->SetDataType(FLOAT32);
    return FromDihip(dihip_dense_pack(stream_of(&ctx), w->GetDataPtr(), n_, k_, DihipDtype(ft_), packed_->GetDataPtr()));
  }
  AsStatus Reshape(RuntimeContext* rt) override {
    AsTensor* h = tensor_map_->at(in_names_[0]).get();
    const Shape& s = h->GetShape();
    if (s.size() != 3 || (int)s[2] != k_ || h->GetDataType() != FLOAT32) return AsStatus::ALLSPARK_PARAM_ERROR;
    batch_ = (int)s[0];
    seq_ = (int)s[1];
    if (rt->is_context && batch_ != 1) return AsStatus::ALLSPARK_PARAM_ERROR;  // the context phase runs one request
    AsTensor* y = tensor_map_->at(out_names_[0]).get();
    y->SetDataType(FLOAT32);
    AS_CHECK_STATUS(y->SetShape(Shape{batch_, 1, n_}));
    return grow_workspace(tensor_map_, dihip_dense_workspace_bytes(std::max(batch_, 1), n_, k_));
  }
  AsStatus Forward(RuntimeContext*) override {
    AsTensor* h = tensor_map_->at(in_names_[0]).get();
    AsTensor* y = tensor_map_->at(out_names_[0]).get();
    AsTensor* wsp = tensor_map_->at("workspace").get();
    // GetLastLine (get_last_line.cpp:81-87): batch rows from row seq - 1 on (context: one request; decoder: seq = 1)
    const float* rows = (const float*)h->GetDataPtr() + (size_t)(seq_ - 1) * k_;
    return FromDihip(dihip_lm_head(stream_of(ctx_), (float*)y->GetDataPtr(), rows, weights_[0]->GetDataPtr(), eps_, packed_->GetDataPtr(), batch_,
                                   n_, k_, wsp->GetDataPtr(), wsp->GetSizeInByte(), sync_->GetDataPtr(), DihipDtype(ft_)));
  }

 private:
  int n_ = 0, k_ = 0, batch_ = 0, seq_ = 0;
  float eps_ = 1e-6f;
  DataType ft_ = BFLOAT16;
  std::unique_ptr<AsTensor> packed_, sync_;
};
REGISTER_OP(DihipLMHead, HIP, DihipLMHeadOp)

// ======================================================================================================== DihipGreedy
// GenerateOp over f32 logits (generate_op.cpp:325-600).  All requests greedy (top_k = 1): ids[b] = argmax, lowest index on ties.
// Any request sampling: dihip_sample with the per-request top_k / top_p / temperature / seed of gen_cfg, the random stream keyed
// by the DEVICE-resident position of the sampled token.  In the decoder phase with device-resident lengths the same launch
// advances "dihip.old_seq_lens" / "dihip.new_seq_lens": the decode step needs nothing from the host and replays as a hipGraph --
// sampling included.
class DihipGreedyOp : public AsOperator {
 public:
  explicit DihipGreedyOp(const std::string& t = "") : AsOperator(t) {}
  AsStatus Init(const OperatorProto& op_proto, const DeviceContext& ctx, const TensorMap& weights_map, TensorMap* tensor_map) override {
    AS_CHECK_STATUS(AsOperator::Init(op_proto, ctx, weights_map, tensor_map));
    tensor_map_->at(out_names_[0])->SetDataType(INT64);
    return AsStatus::ALLSPARK_SUCCESS;
  }
  AsStatus Reshape(RuntimeContext* rt) override {
    AsTensor* x = tensor_map_->at(in_names_[0]).get();
    const Shape& s = x->GetShape();
    // f32 logits from DihipLMHead; FT logits from the tensor-parallel tail (Gemm(splitk) + AllReduce, model_base.py:690-703): cast to f32 first
    if (s.size() < 2 || (x->GetDataType() != FLOAT32 && x->GetDataType() != BFLOAT16 && x->GetDataType() != FLOAT16)) return AsStatus::ALLSPARK_PARAM_ERROR;
    vocab_ = (int)s.back();
    rows_ = (int)(x->Count() / vocab_);
    if (x->GetDataType() != FLOAT32) {
      if (!f32_) f32_ = std::make_unique<AsTensor>(op_name_ + ".logits_f32", DeviceType::HIP, FLOAT32, Shape{(int64_t)rows_ * vocab_});
      AS_CHECK_STATUS(f32_->SetShape(Shape{(int64_t)rows_ * vocab_}));
    }
    // rows of this forward per request (the context phase's prompt length): this operator sees the LAST row's logits only, so the
    // length comes from the graph's input ids (GenerateOpHIP reads it off its [batch, seq, vocab] input) -- used for the sampled
    // token's position when no virtual cache carries the sequence length (sampling_host.h: StagePositions)
    seq_ = 1;
    auto ids = tensor_map_->find(hip_ctx(ctx_).InputIdsName());  // (the runner names the graph's id tensor: a converter export need not call it "input_ids")
    if (rt && rt->is_context && ids != tensor_map_->end() && ids->second->GetShape().size() >= 2) seq_ = std::max(1, (int)ids->second->GetShape()[1]);
    if (rt && rt->GetGenCtxListSize() > 0) {
      AS_CHECK_STATUS(params_.Gather(rt, rows_, stream_of(ctx_)));
      // logits processors / log-probabilities over the requests' DEVICE-resident histories and records (rows form: the step replays)
      std::string why;
      const AsStatus st = proc_.Gather(rt, rows_, vocab_, ctx_->GetModelMaxLength(), true, stream_of(ctx_), &why);
      if (st != AsStatus::ALLSPARK_SUCCESS) {
        std::fprintf(stderr, "[dashinfer_hip] DihipGreedy: %s\n", why.c_str());
        return st;
      }
    }
    AsTensor* y = tensor_map_->at(out_names_[0]).get();
    y->SetDataType(INT64);
    AS_CHECK_STATUS(y->SetShape(Shape{rows_, 1}));
    if (!ws_) ws_ = std::make_unique<AsTensor>(op_name_ + ".argmax_ws", DeviceType::HIP, INT8, Shape{(int64_t)rows_ * 64 * 8 + 256});
    return ws_->SetShape(Shape{(int64_t)rows_ * 64 * 8 + 256});
  }
  AsStatus Forward(RuntimeContext* rt) override {
    AsTensor* x = tensor_map_->at(in_names_[0]).get();
    AsTensor* y = tensor_map_->at(out_names_[0]).get();
    hipStream_t s = stream_of(ctx_);
    const bool on_dev = !rt->is_context && hip_ctx(ctx_).LensOnDevice();
    uint32_t *ca = nullptr, *cb = nullptr;
    if (on_dev) {
      auto o = tensor_map_->find("dihip.old_seq_lens"), n = tensor_map_->find("dihip.new_seq_lens");
      if (o == tensor_map_->end() || n == tensor_map_->end()) return AsStatus::ALLSPARK_INVALID_CALL_ERROR;
      ca = (uint32_t*)o->second->GetDataPtr();
      cb = (uint32_t*)n->second->GetDataPtr();
    }
    const float* logits = (const float*)x->GetDataPtr();
    if (x->GetDataType() != FLOAT32) {
      // GenerateOp's rows: the last row of every request ([batch, seq, vocab] -> seq - 1 in the context phase, batch 1; get_last_line semantics)
      const size_t es = SizeofType(x->GetDataType());
      const char* last = (const char*)x->GetDataPtr() + (x->GetShape().size() == 3 ? (size_t)(x->GetShape()[1] - 1) * vocab_ * es : 0);
      const int rows = x->GetShape().size() == 3 ? (int)x->GetShape()[0] : rows_;
      AS_CHECK_STATUS(FromDihip(dihip_cast_to_f32(s, (float*)f32_->GetDataPtr(), last, (size_t)rows * vocab_, DihipDtype(x->GetDataType()))));
      logits = (const float*)f32_->GetDataPtr();
    }
    const bool extras = proc_.any_processors() || proc_.any_logprobs();
    if (extras && !on_dev && !rt->is_context) {
      std::fprintf(stderr, "[dashinfer_hip] DihipGreedy: logits processors / logprobs in the decoder phase need the device-resident step state\n");
      return AsStatus::ALLSPARK_INVALID_CALL_ERROR;
    }
    if (proc_.any_processors()) {
      // decoder: the step's input id (this operator's own output tensor, which the runner binds as the graph's ids) joins the history at
      // position new_len - 1 and cur_len is the device's new length; context: the history holds the prompt, cur_len = its length
      if (on_dev) {
        AS_CHECK_STATUS(proc_.RunRows(const_cast<float*>(logits), (const int64_t*)y->GetDataPtr(), cb, s));
      } else {
        AS_CHECK_STATUS(proc_.StageCurLen(seq_, s));
        AS_CHECK_STATUS(proc_.RunRows(const_cast<float*>(logits), nullptr, nullptr, s));
      }
    }
    const uint32_t* pos = cb;  // position of the sampled token = tokens in the sequence after this step: new_seq_lens on the device, else staged
    if (!on_dev && (params_.any_sampling() || proc_.any_logprobs())) {
      AS_CHECK_STATUS(params_.StagePositions(rt, rt->is_context ? seq_ : 1, s));
      pos = params_.dev_pos();
    }
    if (params_.any_sampling()) {
      AS_CHECK_STATUS(FromDihip(dihip_sample_rows(s, (int64_t*)y->GetDataPtr(), logits, rows_, vocab_, params_.top_k(), params_.top_p(),
                                                  params_.temperature(), params_.seed(), pos, ca, cb, params_.wide_rows())));
    } else if (on_dev) {
      AS_CHECK_STATUS(FromDihip(dihip_argmax_advance(s, (int64_t*)y->GetDataPtr(), logits, rows_, vocab_, ws_->GetDataPtr(), ws_->GetSizeInByte(), ca, cb)));
    } else {
      AS_CHECK_STATUS(FromDihip(dihip_argmax(s, (int64_t*)y->GetDataPtr(), logits, rows_, vocab_, ws_->GetDataPtr(), ws_->GetSizeInByte())));
    }
    // record index = the sampled token's index in the sequence: the advanced length - 1 on the device, the staged position otherwise
    if (proc_.any_logprobs()) AS_CHECK_STATUS(proc_.LogprobsRows(logits, (const int64_t*)y->GetDataPtr(), pos, on_dev ? -1 : 0, s));
    return AsStatus::ALLSPARK_SUCCESS;
  }

 private:
  int rows_ = 0, vocab_ = 0, seq_ = 1;
  std::unique_ptr<AsTensor> ws_, f32_;
  SamplingParams params_;
  LogitsProcParams proc_;
};
REGISTER_OP(DihipGreedy, HIP, DihipGreedyOp)

}  // namespace allspark
