// moe_op_hip.cpp -- registers op types "MOEA16W8" (class in moe_op_hip.h) and "CalcExpert" for DeviceType::HIP.
#include "moe_op_hip.h"

namespace allspark {

REGISTER_OP(MOEA16W8, HIP, MoeA16W8HIP)

// op type "CalcExpert" (csrc/core/operator/general/calc_expert/calc_expert_op.cpp:14-68): out[t, :] = in[t, :] * expert_weight[t]
// -- the shared expert's output scaled by its sigmoid gate (python/pyhie/allspark/model/qwen_v20_moe.py:366-371).
class CalcExpertHIP : public AsOperator {
 public:
  explicit CalcExpertHIP(const std::string& op_type = "") : AsOperator(op_type) {}

  AsStatus InitV2(const OperatorProto& op_proto, const DeviceContext& ctx, const TensorMap& weights_map, TensorMap& weights_buffer,
                  TensorMap* tensor_map, RuntimeContext* runtime_ctx) override {
    (void)weights_buffer;
    (void)runtime_ctx;
    AS_CHECK_STATUS(AsOperator::Init(op_proto, ctx, weights_map, tensor_map));
    if (ctx.GetDeviceType() != DeviceType::HIP || in_names_.size() != 2) return AsStatus::ALLSPARK_PARAM_ERROR;
    if (op_proto.attr.find("num_experts") == op_proto.attr.end()) return AsStatus::ALLSPARK_PARAM_ERROR;  // calc_expert_op.cpp:26-31
    return AsStatus::ALLSPARK_SUCCESS;
  }

  AsStatus Reshape(RuntimeContext*) override {
    AsTensor* x = tensor_map_->at(in_names_[0]).get();
    Shape shape = x->GetShape();
    if (shape.empty()) return AsStatus::ALLSPARK_PARAM_ERROR;
    hidden_ = (int)shape.back();
    total_token_ = (int)(x->Count() / hidden_);
    if (tensor_map_->at(in_names_[1])->Count() != total_token_) return AsStatus::ALLSPARK_PARAM_ERROR;
    AsTensor* y = tensor_map_->at(out_names_[0]).get();
    y->SetDataType(x->GetDataType());
    return y->SetShape(std::move(shape));
  }

  AsStatus Forward(RuntimeContext*) override {
    AsTensor* x = tensor_map_->at(in_names_[0]).get();
    AsTensor* w = tensor_map_->at(in_names_[1]).get();
    AsTensor* y = tensor_map_->at(out_names_[0]).get();
    if (w->GetDataType() != x->GetDataType()) return AsStatus::ALLSPARK_PARAM_ERROR;
    return FromDihip(dihip_calc_expert(static_cast<const HIPContext*>(ctx_)->GetStream(), y->GetDataPtr(), x->GetDataPtr(), w->GetDataPtr(),
                                       total_token_, hidden_, DihipDtype(x->GetDataType())));
  }

 private:
  int total_token_ = 0, hidden_ = 0;
};

REGISTER_OP(CalcExpert, HIP, CalcExpertHIP)

}  // namespace allspark
