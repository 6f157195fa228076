// span_attn_op_hip.h -- op types "DecOptMHA" / "DecOptMQA" on DeviceType::HIP.
// Host logic mirrors SpanAttnOp (csrc/core/operator/generate_opt/span_attn/span_attn_op.cpp:90-368)
// and SpanAttnOpCUDA (span_attn_op_cuda.cpp:64-392):
//   Init      layer index from the op name, head config from the context, alpha default 1/sqrt(H)
//   Reshape   output [B, seq, n*H]; decode: device arrays for span pointers / lengths sized for batch
//   Forward   prefill (runContext): causal attention over the fused qkv rows (INTERLEAVED) and
//             ContextSpanCopy of K and V into the request's spans;
//             decode (runDecoder): per request span-pointer vectors + old lengths -> device (async
//             H2D, as the reference does per step), DecoderCacheAppend, SpanAttention.
// Input 0 is the fused qkv tensor AFTER the Rotary op, exactly as in the reference graph.
// (class in a header: host/fused_ops_hip.cpp derives the decode-step form DihipRopeSpanAttn from it)
#pragma once
#include <cmath>

#include "dashinfer_hip.h"
#include "operator.h"

namespace allspark {

inline int get_layer_num(const std::string& name) {  // "decoder.layer.12.attention" -> 12 (span_attn_op.cpp:37-55)
  const std::string key = "layer.";
  const size_t p = name.find(key);
  if (p == std::string::npos) return -1;
  size_t q = p + key.size();
  if (q >= name.size() || name[q] < '0' || name[q] > '9') return -1;
  int v = 0;
  while (q < name.size() && name[q] >= '0' && name[q] <= '9') v = v * 10 + (name[q++] - '0');
  return v;
}

class SpanAttnOpHIP : public AsOperator {
 public:
  explicit SpanAttnOpHIP(const std::string& op_type = "") : AsOperator(op_type) {}

  AsStatus Init(const OperatorProto& op_proto, const DeviceContext& ctx, const TensorMap& weights_map,
                TensorMap* tensor_map) override {
    AS_CHECK_STATUS(AsOperator::Init(op_proto, ctx, weights_map, tensor_map));
    if (ctx_->GetCacheSpanSize() <= 0) return AsStatus::ALLSPARK_PARAM_ERROR;
    layer_num_ = get_layer_num(op_name_);
    if (layer_num_ < 0) return AsStatus::ALLSPARK_PARAM_ERROR;
    dtype_ = tensor_map_->at(in_names_[0])->GetDataType();
    tensor_map_->at(out_names_[0])->SetDataType(dtype_);
    auto it = op_proto.attr.find("alpha");
    if (it != op_proto.attr.end()) alpha_ = *(const float*)it->second.c_str();
    // This rank's heads.  g >= nranks: the reference's even split (head_gqa.h:29-49; g % nranks != 0 is its PARAM_ERROR).
    // g < nranks (Qwen2-7B at TP = 8; the reference refuses): each KV head is replicated on nranks / g ranks which divide
    // its query heads, larger shares first -- the rule of dash-infer_amd/tp.py::shard_heads (7 heads on 2 ranks: 4 + 3),
    // which is also how the qkv weight was column-split for this rank.
    {
      const int n = ctx_->GetNumberHeads(), nr = std::max(1, ctx_->GetNranks()), rank = ctx_->GetRank();
      const int g = ctx_->GetNumberGroups() == 0 ? n : ctx_->GetNumberGroups();
      if (n <= 0 || g <= 0 || n % g != 0) return AsStatus::ALLSPARK_PARAM_ERROR;
      if (g >= nr) {
        if (g % nr != 0) return AsStatus::ALLSPARK_PARAM_ERROR;
        g_ = g / nr;
        n_ = n / nr;
      } else {
        if (nr % g != 0) return AsStatus::ALLSPARK_PARAM_ERROR;
        const int rep = nr / g, hpg = n / g, idx = rank % rep;
        g_ = 1;
        n_ = hpg / rep + (idx < hpg % rep ? 1 : 0);
        if (n_ <= 0) return AsStatus::ALLSPARK_PARAM_ERROR;  // more ranks than query heads per KV head
      }
    }
    h_ = ctx_->GetSizePerHead();
    if (alpha_ < 0) alpha_ = 1.0f / std::sqrt((float)h_);
    span_ = ctx_->GetCacheSpanSize();
    kv_mode_ = (int)ctx_->GetCacheMode();
    return AsStatus::ALLSPARK_SUCCESS;
  }

  AsStatus Reshape(RuntimeContext* rt) override {
    AsTensor* x = tensor_map_->at(in_names_[0]).get();
    const Shape& xs = x->GetShape();
    if (xs.size() != 3 || xs[2] != (int64_t)(n_ + 2 * g_) * h_) return AsStatus::ALLSPARK_PARAM_ERROR;
    batch_ = (int)xs[0];
    seq_ = (int)xs[1];
    if (rt->is_context && batch_ != 1) return AsStatus::ALLSPARK_PARAM_ERROR;  // span_attn_op.cpp:127-131
    AS_CHECK_STATUS(tensor_map_->at(out_names_[0])->SetShape(Shape{batch_, seq_, (int64_t)n_ * h_}));
    max_spans_ = (ctx_->GetModelMaxLength() + span_ - 1) / span_;
    const int nb = std::max(batch_, 1);
    auto grow = [&](std::unique_ptr<AsTensor>& t, const char* nm, DeviceType d, DataType dt, int64_t count) {
      if (!t) t = std::make_unique<AsTensor>(op_name_ + nm, d, dt, Shape{count});
      return t->SetShape(Shape{count});
    };
    AS_CHECK_STATUS(grow(k_arr_dev_, ".k_span_array", DeviceType::HIP, POINTER, (int64_t)nb * max_spans_));
    AS_CHECK_STATUS(grow(v_arr_dev_, ".v_span_array", DeviceType::HIP, POINTER, (int64_t)nb * max_spans_));
    AS_CHECK_STATUS(grow(k_arr_host_, ".k_span_array_host", DeviceType::CPU, POINTER, (int64_t)nb * max_spans_));
    AS_CHECK_STATUS(grow(v_arr_host_, ".v_span_array_host", DeviceType::CPU, POINTER, (int64_t)nb * max_spans_));
    AS_CHECK_STATUS(grow(lens_dev_, ".seq_lens", DeviceType::HIP, INT32, 2 * nb));
    AS_CHECK_STATUS(grow(lens_host_, ".seq_lens_host", DeviceType::CPU, INT32, 2 * nb));
    AS_CHECK_STATUS(grow(q_dev_, ".decoder_q", DeviceType::HIP, dtype_, (int64_t)nb * n_ * h_));
    const size_t ws = dihip_span_attn_decode_workspace_bytes(nb, n_, h_, ctx_->GetModelMaxLength(), 0);
    AS_CHECK_STATUS(grow(attn_ws_, ".attn_ws", DeviceType::HIP, INT8, (int64_t)std::max<size_t>(ws, 256)));
    const size_t sb = dihip_span_attn_sync_bytes(nb, n_);
    if (!sync_ || sync_->GetSizeInByte() < sb) {
      sync_ = std::make_unique<AsTensor>(op_name_ + ".sync", DeviceType::HIP, INT8, Shape{(int64_t)sb});
      if (hipMemsetAsync(sync_->GetDataPtr(), 0, sb, Stream()) != hipSuccess) return AsStatus::ALLSPARK_RUNTIME_ERROR;
    }
    return AsStatus::ALLSPARK_SUCCESS;
  }

  // SpanAttnOp::Alloc (span_attn_op.cpp:315-368): before every step each request claims the cache for the tokens this
  // step appends -- seq_ for the request being prefilled, one per request in a decode batch -- through
  // VirtualCache::GetCache(layer, increment), after the reference's sanity check that the cached length equals
  // gen_ctx->step.  The model may call the Alloc of different layers concurrently (CONFIG_CONCURRENT_SPAN,
  // model.cpp:1253-1262): this touches no operator state, only the (thread-safe) cache object of its own layer.
  AsStatus Alloc(RuntimeContext* rt) override {
    auto claim = [&](GenerateContext* gc) -> AsStatus {
      if (!gc || !gc->virtual_k_cache || !gc->virtual_v_cache) return AsStatus::ALLSPARK_PARAM_ERROR;
      if ((size_t)gc->step != gc->virtual_k_cache->GetSeqLength(layer_num_) ||
          (size_t)gc->step != gc->virtual_v_cache->GetSeqLength(layer_num_))
        return AsStatus::ALLSPARK_RUNTIME_ERROR;  // "gen_ctx step and cached seq len mismatch"
      try {
        (void)gc->virtual_k_cache->GetCache(layer_num_, seq_);
        (void)gc->virtual_v_cache->GetCache(layer_num_, seq_);
      } catch (const AsException& e) {
        return e.status();
      }
      return AsStatus::ALLSPARK_SUCCESS;
    };
    if (rt->is_context) return claim(rt->GetContextGenCtx());
    if (rt->GetGenCtxListSize() != batch_) return AsStatus::ALLSPARK_PARAM_ERROR;
    for (int b = 0; b < batch_; ++b) AS_CHECK_STATUS(claim(rt->GetGenCtx(b)));
    return AsStatus::ALLSPARK_SUCCESS;
  }

  AsStatus Forward(RuntimeContext* rt) override { return rt->is_context ? runContext(rt) : runDecoder(rt); }

 protected:
  hipStream_t Stream() const { return static_cast<const HIPContext*>(ctx_)->GetStream(); }

  // this layer's span pointers of one request (the POINTER tensor of VirtualCache::GetCache(layer, 0)) -> staging row b
  AsStatus stageSpans(const GenerateContext* gc, int b, int ntokens) {
    const int need = (ntokens + span_ - 1) / span_;
    if (!gc->virtual_k_cache || !gc->virtual_v_cache || need > max_spans_) return AsStatus::ALLSPARK_EXCEED_LIMIT_ERROR;
    const AsTensor *kt, *vt;
    try {
      kt = &gc->virtual_k_cache->GetCache(layer_num_, 0);
      vt = &gc->virtual_v_cache->GetCache(layer_num_, 0);
    } catch (const AsException& e) {
      return e.status();
    }
    if (kt->Count() < need || vt->Count() < need) return AsStatus::ALLSPARK_EXCEED_LIMIT_ERROR;  // Alloc was not called / failed upstream
    void* const* ks = reinterpret_cast<void* const*>(kt->GetDataPtr());
    void* const* vs = reinterpret_cast<void* const*>(vt->GetDataPtr());
    void** kh = reinterpret_cast<void**>(k_arr_host_->GetDataPtr()) + (size_t)b * max_spans_;
    void** vh = reinterpret_cast<void**>(v_arr_host_->GetDataPtr()) + (size_t)b * max_spans_;
    for (int i = 0; i < need; ++i) {
      kh[i] = ks[i];
      vh[i] = vs[i];
    }
    return AsStatus::ALLSPARK_SUCCESS;
  }
  AsStatus uploadSpans(int nb) {
    const size_t bytes = (size_t)nb * max_spans_ * sizeof(void*);
    if (hipMemcpyAsync(k_arr_dev_->GetDataPtr(), k_arr_host_->GetDataPtr(), bytes, hipMemcpyHostToDevice, Stream()) != hipSuccess ||
        hipMemcpyAsync(v_arr_dev_->GetDataPtr(), v_arr_host_->GetDataPtr(), bytes, hipMemcpyHostToDevice, Stream()) != hipSuccess ||
        hipMemcpyAsync(lens_dev_->GetDataPtr(), lens_host_->GetDataPtr(), (size_t)2 * nb * sizeof(int32_t), hipMemcpyHostToDevice,
                       Stream()) != hipSuccess)
      return AsStatus::ALLSPARK_RUNTIME_ERROR;
    return AsStatus::ALLSPARK_SUCCESS;
  }

  AsStatus runDecoder(RuntimeContext* rt) {
    if (rt->GetGenCtxListSize() != batch_ || seq_ != 1) return AsStatus::ALLSPARK_PARAM_ERROR;
    int32_t* lens = reinterpret_cast<int32_t*>(lens_host_->GetDataPtr());
    for (int b = 0; b < batch_; ++b) {
      const GenerateContext* gc = rt->GetGenCtx(b);
      lens[b] = gc->step;               // old lengths (position of the new token)
      lens[batch_ + b] = gc->step + 1;  // lengths including the new token
      AS_CHECK_STATUS(stageSpans(gc, b, gc->step + 1));
    }
    AS_CHECK_STATUS(uploadSpans(batch_));
    const uint32_t* lens_d = reinterpret_cast<const uint32_t*>(lens_dev_->GetDataPtr());
    void* const* kd = reinterpret_cast<void* const*>(k_arr_dev_->GetDataPtr());
    void* const* vd = reinterpret_cast<void* const*>(v_arr_dev_->GetDataPtr());
    const void* qkv = tensor_map_->at(in_names_[0])->GetDataPtr();
    AS_CHECK_STATUS(FromDihip(dihip_kv_append(Stream(), kd, vd, q_dev_->GetDataPtr(), qkv, lens_d, batch_, n_, g_, h_, span_,
                                              max_spans_, kv_mode_, DihipDtype(dtype_))));
    return FromDihip(dihip_span_attn_decode_sync(Stream(), tensor_map_->at(out_names_[0])->GetDataPtr(), q_dev_->GetDataPtr(),
                                                 (const void* const*)kd, (const void* const*)vd, lens_d + batch_, batch_, n_, g_, h_,
                                                 span_, max_spans_, static_cast<const HIPContext*>(ctx_)->PlanLength(), kv_mode_, DihipDtype(dtype_), alpha_,
                                                 attn_ws_->GetDataPtr(), attn_ws_->GetSizeInByte(), sync_->GetDataPtr(),
                                                 sync_->GetSizeInByte(), DIHIP_ACT_ROWMAJOR));
  }

  AsStatus runContext(RuntimeContext* rt) {
    const GenerateContext* gc = rt->GetContextGenCtx();
    const int stride = (n_ + 2 * g_) * h_;
    if (gc->prefix_len != 0) return runContextWithPrefix(gc, stride);
    const char* qkv = reinterpret_cast<const char*>(tensor_map_->at(in_names_[0])->GetDataPtr());
    const size_t es = SizeofType(dtype_);
    const void* q = qkv;
    const void* k = qkv + (size_t)n_ * h_ * es;
    const void* v = qkv + (size_t)(n_ + g_) * h_ * es;
    AS_CHECK_STATUS(FromDihip(dihip_prefill_attn(Stream(), tensor_map_->at(out_names_[0])->GetDataPtr(), q, k, v, seq_, seq_, stride,
                                                 stride, n_, g_, h_, 1, alpha_, DihipDtype(dtype_))));
    reinterpret_cast<int32_t*>(lens_host_->GetDataPtr())[0] = 0;
    reinterpret_cast<int32_t*>(lens_host_->GetDataPtr())[1] = seq_;
    AS_CHECK_STATUS(stageSpans(gc, 0, seq_));
    AS_CHECK_STATUS(uploadSpans(1));
    AS_CHECK_STATUS(FromDihip(dihip_kv_context_copy(Stream(), reinterpret_cast<void* const*>(k_arr_dev_->GetDataPtr()), k, stride, seq_,
                                                    0, g_, h_, span_, kv_mode_, DihipDtype(dtype_))));
    return FromDihip(dihip_kv_context_copy(Stream(), reinterpret_cast<void* const*>(v_arr_dev_->GetDataPtr()), v, stride, seq_, 0, g_,
                                           h_, span_, kv_mode_, DihipDtype(dtype_)));
  }

  // Prefill over a cached prefix (span_attn_op_cuda.cpp:205-241 copyPrefixSpanToCtxMem, :134-148 UpdateKV,
  // :489-502 MIX format): the prefix is gathered (dequantised) from its spans into the contiguous
  // context workspaces, this step's K/V rows are appended behind it, attention runs with
  // seq_k = prefix + seq over contiguous K/V, and the new rows are written into the spans.
  AsStatus runContextWithPrefix(const GenerateContext* gc, int stride) {
    const int prefix = gc->prefix_len;
    if (prefix % span_ != 0) return AsStatus::ALLSPARK_PARAM_ERROR;  // the prefix cache works in whole spans
    const int total = prefix + seq_;
    const size_t es = SizeofType(dtype_), row = (size_t)g_ * h_ * es;
    auto grow = [&](std::unique_ptr<AsTensor>& t, const char* nm) {
      if (!t) t = std::make_unique<AsTensor>(op_name_ + nm, DeviceType::HIP, dtype_, Shape{(int64_t)total, (int64_t)g_ * h_});
      return t->SetShape(Shape{(int64_t)total, (int64_t)g_ * h_});
    };
    AS_CHECK_STATUS(grow(ctx_k_, ".context_k_workspace"));
    AS_CHECK_STATUS(grow(ctx_v_, ".context_v_workspace"));
    reinterpret_cast<int32_t*>(lens_host_->GetDataPtr())[0] = prefix;
    reinterpret_cast<int32_t*>(lens_host_->GetDataPtr())[1] = total;
    AS_CHECK_STATUS(stageSpans(gc, 0, total));
    AS_CHECK_STATUS(uploadSpans(1));
    void* const* kd = reinterpret_cast<void* const*>(k_arr_dev_->GetDataPtr());
    void* const* vd = reinterpret_cast<void* const*>(v_arr_dev_->GetDataPtr());
    AS_CHECK_STATUS(FromDihip(dihip_kv_prefix_gather(Stream(), ctx_k_->GetDataPtr(), kd, prefix, g_, h_, span_, kv_mode_, DihipDtype(dtype_))));
    AS_CHECK_STATUS(FromDihip(dihip_kv_prefix_gather(Stream(), ctx_v_->GetDataPtr(), vd, prefix, g_, h_, span_, kv_mode_, DihipDtype(dtype_))));
    const char* qkv = reinterpret_cast<const char*>(tensor_map_->at(in_names_[0])->GetDataPtr());
    const void* k = qkv + (size_t)n_ * h_ * es;
    const void* v = qkv + (size_t)(n_ + g_) * h_ * es;
    char* kc = reinterpret_cast<char*>(ctx_k_->GetDataPtr()) + (size_t)prefix * row;
    char* vc = reinterpret_cast<char*>(ctx_v_->GetDataPtr()) + (size_t)prefix * row;
    if (hipMemcpy2DAsync(kc, row, k, (size_t)stride * es, row, seq_, hipMemcpyDeviceToDevice, Stream()) != hipSuccess ||
        hipMemcpy2DAsync(vc, row, v, (size_t)stride * es, row, seq_, hipMemcpyDeviceToDevice, Stream()) != hipSuccess)
      return AsStatus::ALLSPARK_RUNTIME_ERROR;
    AS_CHECK_STATUS(FromDihip(dihip_prefill_attn(Stream(), tensor_map_->at(out_names_[0])->GetDataPtr(), qkv, ctx_k_->GetDataPtr(),
                                                 ctx_v_->GetDataPtr(), seq_, total, stride, g_ * h_, n_, g_, h_, 1, alpha_,
                                                 DihipDtype(dtype_))));
    AS_CHECK_STATUS(FromDihip(dihip_kv_context_copy(Stream(), kd, k, stride, seq_, prefix, g_, h_, span_, kv_mode_, DihipDtype(dtype_))));
    return FromDihip(dihip_kv_context_copy(Stream(), vd, v, stride, seq_, prefix, g_, h_, span_, kv_mode_, DihipDtype(dtype_)));
  }

  std::unique_ptr<AsTensor> ctx_k_, ctx_v_;
  int layer_num_ = -1, n_ = 0, g_ = 0, h_ = 0, span_ = 0, kv_mode_ = 0, batch_ = 0, seq_ = 0, max_spans_ = 0;
  float alpha_ = -1.f;
  DataType dtype_ = BFLOAT16;
  std::unique_ptr<AsTensor> k_arr_dev_, v_arr_dev_, k_arr_host_, v_arr_host_, lens_dev_, lens_host_, q_dev_, attn_ws_, sync_;
};

}  // namespace allspark
