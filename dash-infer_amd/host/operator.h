// operator.h -- AsOperator / OpFactory / REGISTER_OP for the HIP backend.
// Mirrors csrc/core/operator/operator.h:38-201 (same virtuals, same registration macro, same
// factory behaviour: an unknown {op type, device} throws "Unsupported op type.").
#pragma once
#include <functional>
#include <stdexcept>
#include <unordered_map>

#include "as_types.h"

namespace allspark {

class AsException : public std::runtime_error {
 public:
  explicit AsException(const std::string& what, AsStatus st = AsStatus::ALLSPARK_UNKNOWN_ERROR) : std::runtime_error(what), st_(st) {}
  AsStatus status() const { return st_; }  // the reference encodes the status in the message (AS_THROW); kept as a member here

 private:
  AsStatus st_;
};

class AsOperator {
 public:
  explicit AsOperator(const std::string& op_type = "") : op_type_(op_type) {}
  virtual ~AsOperator() = default;
  // the only entry points the model calls (operator.cpp:343-361,529-603)
  AsStatus CallInit(const OperatorProto& op_proto, const DeviceContext& ctx, const TensorMap& weights_map,
                    TensorMap& weights_buffer, TensorMap* tensor_map, RuntimeContext* runtime_ctx);
  AsStatus CallForward(RuntimeContext* runtime_ctx) { return Forward(runtime_ctx); }
  AsStatus CallReshape(RuntimeContext* runtime_ctx) { return Reshape(runtime_ctx); }
  AsStatus CallAlloc(RuntimeContext* runtime_ctx) { return Alloc(runtime_ctx); }
  virtual AsStatus ResetCache() { return AsStatus::ALLSPARK_SUCCESS; }
  std::string GetOpType() const { return op_type_; }
  std::string GetOpName() const { return op_name_; }
  virtual AsStatus InitV2(const OperatorProto& op_proto, const DeviceContext& ctx, const TensorMap& weights_map,
                          TensorMap& weights_buffer, TensorMap* tensor_map, RuntimeContext* runtime_ctx) {
    (void)weights_buffer;
    (void)runtime_ctx;
    return Init(op_proto, ctx, weights_map, tensor_map);
  }

 protected:
  std::string op_type_, op_name_;
  std::vector<std::string> in_names_, out_names_;
  std::vector<AsTensor*> weights_;
  TensorMap* tensor_map_ = nullptr;
  const DeviceContext* ctx_ = nullptr;

  virtual AsStatus Forward() { return AsStatus::ALLSPARK_INVALID_CALL_ERROR; }
  virtual AsStatus Forward(RuntimeContext*) { return Forward(); }
  virtual AsStatus Reshape() { return AsStatus::ALLSPARK_INVALID_CALL_ERROR; }
  virtual AsStatus Reshape(RuntimeContext*) { return Reshape(); }
  virtual AsStatus Alloc(RuntimeContext*) { return AsStatus::ALLSPARK_SUCCESS; }
  // binds in/out tensors by name (creating missing ones) and resolves weights_ (operator.cpp:293-341)
  virtual AsStatus Init(const OperatorProto& op_proto, const DeviceContext& ctx, const TensorMap& weights_map,
                        TensorMap* tensor_map);
};

struct OpRegistType {
  std::string op_type_str;
  DeviceType device_type;
  bool operator==(const OpRegistType& p) const { return op_type_str == p.op_type_str && device_type == p.device_type; }
};
struct OpRegistTypeHashFunction {
  size_t operator()(const OpRegistType& p) const { return std::hash<std::string>{}(p.op_type_str) * 31 + (size_t)p.device_type; }
};
using OpConstructor = std::function<std::unique_ptr<AsOperator>()>;

class OpFactory {
 public:
  static OpFactory& getInstance();
  OpConstructor GetOperator(const OpRegistType& t);  // throws AsException("Unsupported op type.")
  void Register(const OpRegistType& t, OpConstructor c) { op_set_[t] = std::move(c); }

 private:
  std::unordered_map<OpRegistType, OpConstructor, OpRegistTypeHashFunction> op_set_;
};
struct OpRegisterHelper {
  OpRegisterHelper(const OpRegistType& t, OpConstructor c) { OpFactory::getInstance().Register(t, std::move(c)); }
};
#define REGISTER_OP(op_name, device_type, typed_class)                                     \
  static ::allspark::OpRegisterHelper op_name##_##typed_class##Register##_##device_type(   \
      ::allspark::OpRegistType{#op_name, ::allspark::DeviceType::device_type},             \
      []() -> std::unique_ptr<::allspark::AsOperator> { return std::make_unique<typed_class>(#op_name); });

// AsStatus <- status codes of the C-ABI (identical numbering, include/dashinfer_hip.h)
inline AsStatus FromDihip(int rc) { return static_cast<AsStatus>(rc); }
inline int DihipDtype(DataType t) { return t == BFLOAT16 ? 2 : t == FLOAT16 ? 1 : t == FLOAT32 ? 0 : -1; }

}  // namespace allspark
