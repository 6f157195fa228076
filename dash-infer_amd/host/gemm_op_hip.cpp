// gemm_op_hip.cpp -- op type "Gemm" (unquantised 16-bit weights) on DeviceType::HIP: lm_head, the MoE router and every
// layer of an unquantised model (BASELINE configs[0]).  Host logic mirrors GemmOpBase::InitV2 / Reshape and GemmOpGPU::Forward
// (csrc/core/operator/general/gemm/gemm_op.cpp:26-143, gemm_op_gpu.cpp): weights [W [K, N] (, bias [N])]; attributes transB
// (must be 0 here), is_pooler (must be 0), activation (UnaryType), binary_type (ADD = fused residual from the second input;
// under tensor parallelism applied on rank 0 only, gemm_op.cpp:133-137), alpha, splitk (the K-split lm_head of the TP graph,
// model_base.py:690-703: the input row keeps its full width lda = k * nranks and this rank multiplies columns
// [rank * k, (rank + 1) * k) of it with its row block of the weight, gemm_op.cpp:95-98).  The weight is re-laid-out once at
// InitV2 into dihip tile-major order (dihip_dense_pack); Forward only enqueues dihip_gemm_a16w16.
#include <algorithm>

#include "dashinfer_hip.h"
#include "operator.h"

namespace allspark {

class GemmHIP : public AsOperator {
 public:
  explicit GemmHIP(const std::string& t = "") : AsOperator(t) {}

  AsStatus InitV2(const OperatorProto& op_proto, const DeviceContext& ctx, const TensorMap& weights_map, TensorMap& weights_buffer,
                  TensorMap* tensor_map, RuntimeContext* runtime_ctx) override {
    (void)weights_buffer;
    (void)runtime_ctx;
    AS_CHECK_STATUS(AsOperator::Init(op_proto, ctx, weights_map, tensor_map));
    if (ctx.GetDeviceType() != DeviceType::HIP) return AsStatus::ALLSPARK_PARAM_ERROR;
    if (weights_.size() != 1 && weights_.size() != 2) return AsStatus::ALLSPARK_PARAM_ERROR;  // gemm_op.cpp:31-36
    auto get = [&](const char* k) -> const char* {
      auto it = op_proto.attr.find(k);
      return it == op_proto.attr.end() ? nullptr : it->second.c_str();
    };
    if (const char* p = get("transB"))
      if (*(const bool*)p) return AsStatus::ALLSPARK_PARAM_ERROR;
    if (const char* p = get("is_pooler"))
      if (*(const bool*)p) return AsStatus::ALLSPARK_PARAM_ERROR;
    if (const char* p = get("activation")) activation_ = *(const UnaryType*)p;
    if (const char* p = get("binary_type")) binary_type_ = *(const int*)p;
    if (const char* p = get("alpha")) alpha_ = *(const float*)p;
    if (const char* p = get("splitk")) split_k_ = *(const bool*)p;
    if (binary_type_ != 0 && binary_type_ != 1) return AsStatus::ALLSPARK_PARAM_ERROR;  // ADD only (as the decoder graphs use it)
    const AsTensor* w = weights_[0];
    if (w->GetShape().size() != 2) return AsStatus::ALLSPARK_PARAM_ERROR;
    k_ = (int)w->GetShape()[0];
    n_ = (int)w->GetShape()[1];
    ftype_ = w->GetDataType();
    if (ftype_ != FLOAT16 && ftype_ != BFLOAT16) return AsStatus::ALLSPARK_PARAM_ERROR;
    nranks_ = std::max(1, ctx.GetNranks());
    rank_ = ctx.GetRank();
    lda_ = split_k_ ? k_ * nranks_ : k_;
    tensor_map_->at(out_names_[0])->SetDataType(ftype_);
    const HIPContext& hctx = static_cast<const HIPContext&>(ctx);
    packed_w_ = std::make_unique<AsTensor>(op_name_ + ".packed_w", DeviceType::HIP, INT8, Shape{(int64_t)dihip_dense_packed_weight_bytes(n_, k_)});
    if (!packed_w_->GetDataPtr()) return AsStatus::ALLSPARK_MEMORY_ERROR;
    sync_ = std::make_unique<AsTensor>(op_name_ + ".sync", DeviceType::HIP, INT8, Shape{(int64_t)dihip_gemm_lowp_sync_bytes()});
    if (hipMemsetAsync(sync_->GetDataPtr(), 0, sync_->GetSizeInByte(), hctx.GetStream()) != hipSuccess) return AsStatus::ALLSPARK_RUNTIME_ERROR;
    return FromDihip(dihip_dense_pack(hctx.GetStream(), w->GetDataPtr(), n_, k_, DihipDtype(ftype_), packed_w_->GetDataPtr()));
  }

  AsStatus Reshape(RuntimeContext*) override {
    AsTensor* x = tensor_map_->at(in_names_[0]).get();
    Shape ys = x->GetShape();
    if (ys.empty() || (int)ys.back() != lda_) return AsStatus::ALLSPARK_PARAM_ERROR;
    m_ = (int)(x->Count() / lda_);
    ys.back() = n_;
    AsTensor* y = tensor_map_->at(out_names_[0]).get();
    y->SetDataType(x->GetDataType());
    AS_CHECK_STATUS(y->SetShape(std::move(ys)));
    int64_t ws = (int64_t)dihip_dense_workspace_bytes(std::max(m_, 1), n_, k_);
    if (split_k_) ws += (int64_t)std::max(m_, 1) * k_ * 2 + 256;  // the contiguous copy of this rank's K slice of the rows
    AsTensor* wsp = tensor_map_->at("workspace").get();
    if (wsp->GetSizeInByte() < (size_t)ws) AS_CHECK_STATUS(wsp->SetShape(Shape{ws}));
    return AsStatus::ALLSPARK_SUCCESS;
  }

  AsStatus Forward(RuntimeContext*) override {
    AsTensor* x = tensor_map_->at(in_names_[0]).get();
    AsTensor* y = tensor_map_->at(out_names_[0]).get();
    AsTensor* wsp = tensor_map_->at("workspace").get();
    if (x->GetDataType() != ftype_) return AsStatus::ALLSPARK_PARAM_ERROR;
    if (m_ == 0) return AsStatus::ALLSPARK_SUCCESS;
    hipStream_t s = static_cast<const HIPContext*>(ctx_)->GetStream();
    const void* bias = weights_.size() == 2 ? weights_[1]->GetDataPtr() : nullptr;
    const bool add_residual = binary_type_ == 1 && in_names_.size() > 1 && (nranks_ <= 1 || rank_ == 0);  // gemm_op.cpp:133-137
    const void* residual = add_residual ? tensor_map_->at(in_names_[1])->GetDataPtr() : nullptr;
    const void* xin = x->GetDataPtr();
    char* ws = (char*)wsp->GetDataPtr();
    size_t ws_bytes = wsp->GetSizeInByte();
    if (split_k_ && nranks_ > 1) {
      // columns [rank * k, (rank + 1) * k) of rows of width lda: a strided copy into the head of the workspace
      const size_t slice = ((size_t)m_ * k_ * 2 + 255) & ~(size_t)255;
      if (hipMemcpy2DAsync(ws, (size_t)k_ * 2, (const char*)xin + (size_t)rank_ * k_ * 2, (size_t)lda_ * 2, (size_t)k_ * 2, (size_t)m_,
                           hipMemcpyDeviceToDevice, s) != hipSuccess)
        return AsStatus::ALLSPARK_RUNTIME_ERROR;
      xin = ws;
      ws += slice;
      ws_bytes -= slice;
    }
    return FromDihip(dihip_gemm_a16w16(s, xin, packed_w_->GetDataPtr(), bias, residual, y->GetDataPtr(), m_, n_, k_, (int)activation_, alpha_,
                                       ws, ws_bytes, sync_->GetDataPtr(), DihipDtype(ftype_)));
  }

 private:
  int m_ = 0, n_ = 0, k_ = 0, lda_ = 0, nranks_ = 1, rank_ = 0, binary_type_ = 0;
  bool split_k_ = false;
  float alpha_ = 1.0f;
  UnaryType activation_ = UNARYTYPE_UNDEFINED;
  DataType ftype_ = BFLOAT16;
  std::unique_ptr<AsTensor> packed_w_, sync_;
};
REGISTER_OP(Gemm, HIP, GemmHIP)

}  // namespace allspark
