#include "operator.h"

namespace allspark {

OpFactory& OpFactory::getInstance() {
  static OpFactory f;
  return f;
}
OpConstructor OpFactory::GetOperator(const OpRegistType& t) {
  auto it = op_set_.find(t);
  if (it == op_set_.end()) throw AsException("Unsupported op type.");  // operator.cpp:379-386
  return it->second;
}

AsStatus AsOperator::Init(const OperatorProto& op_proto, const DeviceContext& ctx, const TensorMap& weights_map,
                          TensorMap* tensor_map) {
  tensor_map_ = tensor_map;
  ctx_ = &ctx;
  op_name_ = op_proto.op_name;
  in_names_.clear();
  out_names_.clear();
  weights_.clear();
  for (const auto& n : op_proto.inputs) {
    if (tensor_map_->count(n) == 0) tensor_map_->emplace(n, std::make_shared<AsTensor>(n, ctx.GetDeviceType(), FLOAT32));
    in_names_.push_back(n);
  }
  for (const auto& n : op_proto.outputs) {
    if (tensor_map_->count(n) == 0) tensor_map_->emplace(n, std::make_shared<AsTensor>(n, ctx.GetDeviceType(), FLOAT32));
    out_names_.push_back(n);
  }
  for (const auto& n : op_proto.weights) {
    auto it = weights_map.find(n);
    if (it == weights_map.end()) return AsStatus::ALLSPARK_PARAM_ERROR;
    weights_.push_back(it->second.get());
  }
  return AsStatus::ALLSPARK_SUCCESS;
}

AsStatus AsOperator::CallInit(const OperatorProto& op_proto, const DeviceContext& ctx, const TensorMap& weights_map,
                              TensorMap& weights_buffer, TensorMap* tensor_map, RuntimeContext* runtime_ctx) {
  return InitV2(op_proto, ctx, weights_map, weights_buffer, tensor_map, runtime_ctx);
}

}  // namespace allspark
