// allgather_op_hip.cpp -- op type "AllGather" on DeviceType::HIP (AllGatherOp,
// csrc/core/operator/nccl/allgather/allgather_op.cpp:27-58,134-163): every rank contributes [rows, n] and receives
// [rows, nranks * n] -- the rank-major result of ncclAllGather is transposed into row-major (transpose_axis_01), as the
// reference does; the tensor-parallel embedding (VSPLIT table) is its user in the Qwen2 graph (qwen_v15.py:193-200).
// Reshape grows the shared "workspace" to the gathered size (allgather_op.cpp:147-150).  Only enqueues (the reference
// brackets the collective with two host synchronisations, :39 and :53).
#include "dashinfer_hip.h"
#include "operator.h"

namespace allspark {

class AllGatherOpHIP : public AsOperator {
 public:
  explicit AllGatherOpHIP(const std::string& op_type = "") : AsOperator(op_type) {}
  AsStatus Init(const OperatorProto& op_proto, const DeviceContext& ctx, const TensorMap& weights_map,
                TensorMap* tensor_map) override {
    AS_CHECK_STATUS(AsOperator::Init(op_proto, ctx, weights_map, tensor_map));
    if (ctx.GetDeviceType() != DeviceType::HIP) return AsStatus::ALLSPARK_RUNTIME_ERROR;
    tensor_map_->at(out_names_[0])->SetDataType(tensor_map_->at(in_names_[0])->GetDataType());
    return AsStatus::ALLSPARK_SUCCESS;
  }
  AsStatus Reshape(RuntimeContext*) override {
    AsTensor* x = tensor_map_->at(in_names_[0]).get();
    Shape out = x->GetShape();
    if (out.empty()) return AsStatus::ALLSPARK_PARAM_ERROR;
    n_ = out.back();
    m_ = x->Count() / std::max<int64_t>(n_, 1);
    out.back() *= ctx_->GetNranks();
    AsTensor* y = tensor_map_->at(out_names_[0]).get();
    y->SetDataType(x->GetDataType());
    const int64_t bytes = m_ * n_ * ctx_->GetNranks() * (int64_t)SizeofType(x->GetDataType());
    AS_CHECK_STATUS(y->SetShape(std::move(out)));
    AsTensor* ws = tensor_map_->at("workspace").get();
    if ((int64_t)ws->GetSizeInByte() < bytes) AS_CHECK_STATUS(ws->SetShape(Shape{bytes}));
    return AsStatus::ALLSPARK_SUCCESS;
  }
  AsStatus Forward(RuntimeContext*) override {
    const HIPContext* h = static_cast<const HIPContext*>(ctx_);
    AsTensor* x = tensor_map_->at(in_names_[0]).get();
    AsTensor* y = tensor_map_->at(out_names_[0]).get();
    if (h->GetNranks() > 1 && !h->GetRCCLComm()) return AsStatus::ALLSPARK_PARAM_ERROR;
    return FromDihip(dihip_allgather_rows(h->GetRCCLComm(), h->GetStream(), x->GetDataPtr(), tensor_map_->at("workspace")->GetDataPtr(),
                                          y->GetDataPtr(), (int)m_, (size_t)n_ * SizeofType(x->GetDataType()), h->GetNranks()));
  }

 private:
  int64_t m_ = 0, n_ = 0;
};
REGISTER_OP(AllGather, HIP, AllGatherOpHIP)

}  // namespace allspark
