// moe_op_hip.h -- class of op type "MOEA16W8" on DeviceType::HIP (registered in moe_op_hip.cpp); a header because the fused
// MoE block operator (fused_ops_hip.cpp: DihipMoeBlock) derives from it for the expert re-layout.
// op type "MOEA16W8" on DeviceType::HIP: the reference's MOE operator with weight-only int8 experts.
// Host logic mirrors MoeOp::{Init,Reshape,Forward} (csrc/core/operator/general/moe/moe_op.cpp:60-210,212-336,338-460):
//   inputs  [hidden rows [T(,1), hidden], router logits [T(,1), num_experts]]          (in_names_[0], in_names_[1])
//   weights stacked over the experts, the A16W8 triple per projection (gemm_a16w8.cpp:26-31 order W, scales, zeros):
//           gate_up_proj  int8 [E, hidden, 2*proj]  + scales / zeros FT [E, G, 2*proj]   (columns [gate | up], unary.cu:122-132)
//           down_proj     int8 [E, proj, hidden]    + scales / zeros FT [E, G, hidden]
//   attrs   num_experts, num_experts_per_tok (moe_op.cpp:63-80), GroupSize (optional, gemm_a16w8.cpp:66-74), use_ep (optional,
//           moe_op.cpp:103-117: rank r holds experts [r * E / nranks, (r + 1) * E / nranks) -- the stacks then have E / nranks
//           entries -- and leaves the other experts' terms to the AllReduce that follows the operator)
//   output  [T(,1), hidden]
// InitV2 re-lays every expert out once (column halves of gate_up become the gate and the up stack of dihip tile-major
// tensors); Reshape sizes the routing tensors and grows the shared "workspace"; Forward only enqueues on the context
// stream: route -> expert GEMVs over (token, expert) slots -> combine (include/dashinfer_hip.h section 1b).  The
// reference's reorder / pad / batched-GEMM tensors (experts_idx_, experts_seq_, *_array_ptr ...) have no counterpart.
#pragma once
#include <algorithm>

#include "dashinfer_hip.h"
#include "operator.h"

namespace allspark {

class MoeA16W8HIP : public AsOperator {
 public:
  explicit MoeA16W8HIP(const std::string& op_type = "") : AsOperator(op_type) {}

  AsStatus InitV2(const OperatorProto& op_proto, const DeviceContext& ctx, const TensorMap& weights_map,
                  TensorMap& weights_buffer, TensorMap* tensor_map, RuntimeContext* runtime_ctx) override {
    (void)weights_buffer;
    (void)runtime_ctx;
    AS_CHECK_STATUS(AsOperator::Init(op_proto, ctx, weights_map, tensor_map));
    if (ctx.GetDeviceType() != DeviceType::HIP) return AsStatus::ALLSPARK_PARAM_ERROR;
    if (weights_.size() != 6 || in_names_.size() != 2) return AsStatus::ALLSPARK_PARAM_ERROR;
    AS_CHECK_STATUS(ParseMoeAttrs(op_proto, ctx));
    return PackExperts(weights_.data(), ctx);
  }

 protected:
  // attributes of MoeOp::Init (moe_op.cpp:63-117)
  AsStatus ParseMoeAttrs(const OperatorProto& op_proto, const DeviceContext& ctx) {
    const auto& attr = op_proto.attr;
    auto geti = [&](const char* k, int* v) {
      auto it = attr.find(k);
      if (it == attr.end()) return false;
      *v = *(const int*)it->second.c_str();
      return true;
    };
    if (!geti("num_experts", &num_expert_) || num_expert_ <= 0 || num_expert_ > 256) return AsStatus::ALLSPARK_PARAM_ERROR;  // :65-74
    if (!geti("num_experts_per_tok", &top_k_) || top_k_ <= 0 || top_k_ > num_expert_) return AsStatus::ALLSPARK_PARAM_ERROR;  // :75-80
    geti("GroupSize", &group_size_);
    ep_num_ = num_expert_;
    ep_first_ = 0;
    if (attr.find("use_ep") != attr.end()) {  // moe_op.cpp:103-108
      const int nranks = std::max(1, ctx.GetNranks());
      if (num_expert_ % nranks) return AsStatus::ALLSPARK_PARAM_ERROR;
      ep_num_ = num_expert_ / nranks;
      ep_first_ = ctx.GetRank() * ep_num_;
    }
    return AsStatus::ALLSPARK_SUCCESS;
  }

  // w[0..5] = gate_up_proj W / scales / zeros, down_proj W / scales / zeros, stacked over this rank's experts
  AsStatus PackExperts(AsTensor* const* w, const DeviceContext& ctx) {
    const AsTensor* gu = w[0];
    const AsTensor* dn = w[3];
    if (gu->GetShape().size() != 3 || dn->GetShape().size() != 3 || gu->GetDataType() != INT8 || dn->GetDataType() != INT8)
      return AsStatus::ALLSPARK_PARAM_ERROR;
    if ((int)gu->GetShape()[0] != ep_num_ || (int)dn->GetShape()[0] != ep_num_) return AsStatus::ALLSPARK_PARAM_ERROR;
    hidden_ = (int)gu->GetShape()[1];       // moe_op.cpp:120-121
    proj_ = (int)gu->GetShape()[2] / 2;
    if ((int)dn->GetShape()[1] != proj_ || (int)dn->GetShape()[2] != hidden_) return AsStatus::ALLSPARK_PARAM_ERROR;
    ftype_ = w[1]->GetDataType();
    if (ftype_ != BFLOAT16) return AsStatus::ALLSPARK_PARAM_ERROR;  // the expert GEMVs are bf16
    const HIPContext& hctx = static_cast<const HIPContext&>(ctx);
    hipStream_t s = hctx.GetStream();
    const int G = (int)w[1]->GetShape()[1];
    const size_t wb_gu = dihip_gemm_lowp_packed_weight_bytes(8, proj_, hidden_), sb_gu = dihip_gemm_lowp_packed_sz_bytes(proj_, hidden_, group_size_);
    const size_t wb_dn = dihip_gemm_lowp_packed_weight_bytes(8, hidden_, proj_), sb_dn = dihip_gemm_lowp_packed_sz_bytes(hidden_, proj_, group_size_);
    auto mk = [&](const char* n, size_t bytes) { return std::make_unique<AsTensor>(op_name_ + n, DeviceType::HIP, INT8, Shape{(int64_t)bytes}); };
    gate_w_ = mk(".gate_w", wb_gu * ep_num_);
    up_w_ = mk(".up_w", wb_gu * ep_num_);
    down_w_ = mk(".down_w", wb_dn * ep_num_);
    gate_sz_ = mk(".gate_sz", sb_gu * ep_num_);
    up_sz_ = mk(".up_sz", sb_gu * ep_num_);
    down_sz_ = mk(".down_sz", sb_dn * ep_num_);
    // staging for one expert's column halves: W [hidden, proj] int8, scales / zeros [G, proj] FT
    auto tmp_w = mk(".tmp_w", (size_t)hidden_ * proj_), tmp_s = mk(".tmp_s", (size_t)G * proj_ * 2), tmp_z = mk(".tmp_z", (size_t)G * proj_ * 2);
    for (auto* t : {gate_w_.get(), up_w_.get(), down_w_.get(), gate_sz_.get(), up_sz_.get(), down_sz_.get(), tmp_w.get(), tmp_s.get(), tmp_z.get()})
      if (!t->GetDataPtr()) return AsStatus::ALLSPARK_MEMORY_ERROR;
    const char* guw = (const char*)w[0]->GetDataPtr();
    const char* gus = (const char*)w[1]->GetDataPtr();
    const char* guz = (const char*)w[2]->GetDataPtr();
    for (int e = 0; e < ep_num_; ++e) {
      for (int half = 0; half < 2; ++half) {  // columns [0, proj) = gate, [proj, 2 proj) = up
        const char* w0 = guw + ((size_t)e * hidden_ * 2 * proj_) + (size_t)half * proj_;
        const char* s0 = gus + ((size_t)e * G * 2 * proj_ + (size_t)half * proj_) * 2;
        const char* z0 = guz + ((size_t)e * G * 2 * proj_ + (size_t)half * proj_) * 2;
        if (hipMemcpy2DAsync(tmp_w->GetDataPtr(), proj_, w0, (size_t)2 * proj_, proj_, hidden_, hipMemcpyDeviceToDevice, s) != hipSuccess ||
            hipMemcpy2DAsync(tmp_s->GetDataPtr(), (size_t)proj_ * 2, s0, (size_t)4 * proj_, (size_t)proj_ * 2, G, hipMemcpyDeviceToDevice, s) != hipSuccess ||
            hipMemcpy2DAsync(tmp_z->GetDataPtr(), (size_t)proj_ * 2, z0, (size_t)4 * proj_, (size_t)proj_ * 2, G, hipMemcpyDeviceToDevice, s) != hipSuccess)
          return AsStatus::ALLSPARK_RUNTIME_ERROR;
        AsTensor* dw = half ? up_w_.get() : gate_w_.get();
        AsTensor* ds = half ? up_sz_.get() : gate_sz_.get();
        AS_CHECK_STATUS(FromDihip(dihip_gemm_lowp_pack(s, 8, tmp_w->GetDataPtr(), tmp_s->GetDataPtr(), tmp_z->GetDataPtr(), proj_, hidden_,
                                                       group_size_, DihipDtype(ftype_), (char*)dw->GetDataPtr() + (size_t)e * wb_gu,
                                                       (char*)ds->GetDataPtr() + (size_t)e * sb_gu)));
      }
      const int Gd = (int)w[4]->GetShape()[1];
      AS_CHECK_STATUS(FromDihip(dihip_gemm_lowp_pack(
          s, 8, (const char*)w[3]->GetDataPtr() + (size_t)e * proj_ * hidden_, (const char*)w[4]->GetDataPtr() + (size_t)e * Gd * hidden_ * 2,
          (const char*)w[5]->GetDataPtr() + (size_t)e * Gd * hidden_ * 2, hidden_, proj_, group_size_, DihipDtype(ftype_),
          (char*)down_w_->GetDataPtr() + (size_t)e * wb_dn, (char*)down_sz_->GetDataPtr() + (size_t)e * sb_dn)));
    }
    if (hipStreamSynchronize(s) != hipSuccess) return AsStatus::ALLSPARK_RUNTIME_ERROR;  // the staging tensors die here
    return AsStatus::ALLSPARK_SUCCESS;
  }

 public:
  AsStatus Reshape(RuntimeContext*) override {
    AsTensor* x = tensor_map_->at(in_names_[0]).get();
    AsTensor* lg = tensor_map_->at(in_names_[1]).get();
    Shape oshape = x->GetShape();
    if (oshape.empty() || (int)oshape.back() != hidden_) return AsStatus::ALLSPARK_PARAM_ERROR;
    total_token_ = (int)(x->Count() / hidden_);
    if (lg->Count() != (int64_t)total_token_ * num_expert_) return AsStatus::ALLSPARK_PARAM_ERROR;
    AsTensor* y = tensor_map_->at(out_names_[0]).get();
    y->SetDataType(x->GetDataType());
    AS_CHECK_STATUS(y->SetShape(std::move(oshape)));
    const int64_t slots = (int64_t)std::max(total_token_, 1) * top_k_;
    if (!experts_score_ || experts_score_->Count() < slots) {  // moe_op.cpp:221-232 (experts_score_, topk_indice_)
      experts_score_ = std::make_unique<AsTensor>(op_name_ + ".experts_score", DeviceType::HIP, FLOAT32, Shape{slots});
      topk_indice_ = std::make_unique<AsTensor>(op_name_ + ".topk_indice", DeviceType::HIP, INT32, Shape{slots});
      if (!experts_score_->GetDataPtr() || !topk_indice_->GetDataPtr()) return AsStatus::ALLSPARK_MEMORY_ERROR;
    }
    const int64_t ws = (int64_t)dihip_moe_workspace_bytes(std::max(total_token_, 1), top_k_, hidden_, proj_);
    AsTensor* wsp = tensor_map_->at("workspace").get();  // shared scratch, grows only (model.cpp:241-243)
    if (wsp->GetSizeInByte() < (size_t)ws) AS_CHECK_STATUS(wsp->SetShape(Shape{ws}));
    return AsStatus::ALLSPARK_SUCCESS;
  }

  AsStatus Forward(RuntimeContext*) override {
    AsTensor* x = tensor_map_->at(in_names_[0]).get();
    AsTensor* lg = tensor_map_->at(in_names_[1]).get();
    AsTensor* y = tensor_map_->at(out_names_[0]).get();
    AsTensor* wsp = tensor_map_->at("workspace").get();
    if (x->GetDataType() != ftype_) return AsStatus::ALLSPARK_PARAM_ERROR;
    hipStream_t s = static_cast<const HIPContext*>(ctx_)->GetStream();
    // decode batches: routing and the same-expert slot grouping in one launch (dihip_moe_route_grouped); the operator has one
    // output tensor, so the finalize-routing launch stays (the fused decode step folds it into its combine, dihip_moe_combine)
    const bool one_launch_route = total_token_ > 1 && (int64_t)total_token_ * top_k_ <= 2048 && lg->GetDataType() != FLOAT32;
    if (one_launch_route) {
      AS_CHECK_STATUS(FromDihip(dihip_moe_route_grouped(s, lg->GetDataPtr(), total_token_, num_expert_, top_k_, (float*)experts_score_->GetDataPtr(),
                                                        (int32_t*)topk_indice_->GetDataPtr(), DihipDtype(lg->GetDataType()), ep_first_, ep_num_,
                                                        hidden_, proj_, wsp->GetDataPtr(), wsp->GetSizeInByte())));
    } else {
      AS_CHECK_STATUS(FromDihip(dihip_moe_route_ep(s, lg->GetDataPtr(), total_token_, num_expert_, top_k_, (float*)experts_score_->GetDataPtr(),
                                                   (int32_t*)topk_indice_->GetDataPtr(), DihipDtype(lg->GetDataType()), ep_first_, ep_num_)));
    }
    return FromDihip(dihip_moe_experts_ex(s, 8, x->GetDataPtr(), (const int32_t*)topk_indice_->GetDataPtr(), (const float*)experts_score_->GetDataPtr(),
                                          gate_w_->GetDataPtr(), gate_sz_->GetDataPtr(), up_w_->GetDataPtr(), up_sz_->GetDataPtr(),
                                          down_w_->GetDataPtr(), down_sz_->GetDataPtr(), total_token_, top_k_, hidden_, proj_, group_size_,
                                          y->GetDataPtr(), wsp->GetDataPtr(), wsp->GetSizeInByte(), DihipDtype(ftype_),
                                          one_launch_route ? DIHIP_MOE_PREGROUPED : 0));
  }

 protected:
  int num_expert_ = 0, top_k_ = 0, hidden_ = 0, proj_ = 0, group_size_ = -1, total_token_ = 0, ep_num_ = 0, ep_first_ = 0;
  DataType ftype_ = BFLOAT16;
  std::unique_ptr<AsTensor> gate_w_, up_w_, down_w_, gate_sz_, up_sz_, down_sz_, experts_score_, topk_indice_;
};

}  // namespace allspark
