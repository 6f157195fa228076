// gemm_lowp_op_hip.cpp -- op types "GemmA16W8" / "GemmA16W4" on DeviceType::HIP.
// Host logic mirrors GemmA16W8Base::InitV2 / GemmA16W8GPU::{InitV2,Reshape,Forward}
// (csrc/core/operator/general/gemm_lowp/gemm_a16w8.cpp:21-125, gemm_a16w8_gpu.cpp:30-267) and
// GemmA16W4GPU (gemm_a16w4_gpu.cpp:26-230): weights [W, scales, zeros, (bias)]; attributes alpha,
// activation, GroupSize, transB (must be 0), is_pooler (must be 0); the weight is re-laid-out
// once at InitV2 (the reference's N32K16 reorder becomes dihip tile-major packing); Reshape sizes
// the output and grows the shared "workspace" tensor; Forward only enqueues on the context stream.
// All device work goes through the C-ABI (include/dashinfer_hip.h).
#include <algorithm>

#include "dashinfer_hip.h"
#include "operator.h"

namespace allspark {

template <int WBITS>
class GemmLowpHIP : public AsOperator {
 public:
  explicit GemmLowpHIP(const std::string& op_type = "") : AsOperator(op_type) {}

  AsStatus InitV2(const OperatorProto& op_proto, const DeviceContext& ctx, const TensorMap& weights_map,
                  TensorMap& weights_buffer, TensorMap* tensor_map, RuntimeContext* runtime_ctx) override {
    (void)weights_buffer;
    (void)runtime_ctx;
    AS_CHECK_STATUS(AsOperator::Init(op_proto, ctx, weights_map, tensor_map));
    if (ctx.GetDeviceType() != DeviceType::HIP) return AsStatus::ALLSPARK_PARAM_ERROR;
    if (weights_.size() != 3 && weights_.size() != 4) return AsStatus::ALLSPARK_PARAM_ERROR;  // gemm_a16w8.cpp:26-31
    const auto& attr = op_proto.attr;
    auto get = [&](const char* k) -> const char* {
      auto it = attr.find(k);
      return it == attr.end() ? nullptr : it->second.c_str();
    };
    if (const char* p = get("transB")) {
      if (*(const bool*)p) return AsStatus::ALLSPARK_PARAM_ERROR;  // :75-78
    }
    if (const char* p = get("is_pooler")) {
      if (*(const bool*)p) return AsStatus::ALLSPARK_PARAM_ERROR;  // :80-83
    }
    if (const char* p = get("activation")) activation_ = *(const UnaryType*)p;
    if (const char* p = get("alpha")) alpha_ = *(const float*)p;
    if (const char* p = get("GroupSize")) group_size_ = *(const int*)p;
    if (group_size_ != -1 && (group_size_ < 32 || group_size_ % 32 != 0)) return AsStatus::ALLSPARK_PARAM_ERROR;
    const AsTensor* w = weights_[0];
    if (w->GetShape().size() != 2) return AsStatus::ALLSPARK_PARAM_ERROR;
    k_ = (int)w->GetShape()[0];
    // A16W4: the stored weight is [K, ceil(N/2)] bytes; N comes from the scales (gemm_a16w4.cpp:60-75)
    n_ = (int)weights_[1]->GetShape().back();
    if (WBITS == 8 && (w->GetDataType() != INT8 || (int)w->GetShape()[1] != n_)) return AsStatus::ALLSPARK_PARAM_ERROR;
    if (WBITS == 4 && (w->GetDataType() != UINT8 || (int)w->GetShape()[1] != (n_ + 1) / 2)) return AsStatus::ALLSPARK_PARAM_ERROR;
    ftype_ = weights_[1]->GetDataType();
    if (ftype_ != FLOAT16 && ftype_ != BFLOAT16) return AsStatus::ALLSPARK_PARAM_ERROR;
    tensor_map_->at(out_names_[0])->SetDataType(ftype_);  // type inference at Init: the next operator's Init reads it
    // re-layout once (gemm_a16w8_gpu.cpp:421-473 does the same job for the CUDA kernels)
    const HIPContext& hctx = static_cast<const HIPContext&>(ctx);
    packed_w_ = std::make_unique<AsTensor>(op_name_ + ".packed_w", DeviceType::HIP, INT8,
                                           Shape{(int64_t)dihip_gemm_lowp_packed_weight_bytes(WBITS, n_, k_)});
    packed_sz_ = std::make_unique<AsTensor>(op_name_ + ".packed_sz", DeviceType::HIP, INT8,
                                            Shape{(int64_t)dihip_gemm_lowp_packed_sz_bytes(n_, k_, group_size_)});
    if (!packed_w_->GetDataPtr() || !packed_sz_->GetDataPtr()) return AsStatus::ALLSPARK_MEMORY_ERROR;
    sync_ = std::make_unique<AsTensor>(op_name_ + ".sync", DeviceType::HIP, INT8, Shape{(int64_t)dihip_gemm_lowp_sync_bytes()});
    if (hipMemsetAsync(sync_->GetDataPtr(), 0, sync_->GetSizeInByte(), hctx.GetStream()) != hipSuccess)
      return AsStatus::ALLSPARK_RUNTIME_ERROR;
    AS_CHECK_STATUS(FromDihip(dihip_gemm_lowp_pack(hctx.GetStream(), WBITS, weights_[0]->GetDataPtr(), weights_[1]->GetDataPtr(),
                                                   weights_[2]->GetDataPtr(), n_, k_, group_size_, DihipDtype(ftype_),
                                                   packed_w_->GetDataPtr(), packed_sz_->GetDataPtr())));
    // (the reference re-lays-out in place, gemm_a16w8_gpu.cpp:456-469; here the unpacked source is released once packed, when the
    // weight map owns it -- no weight is held twice)
    if (weights_[0]->OwnsStorage()) {
      if (hipStreamSynchronize(hctx.GetStream()) != hipSuccess) return AsStatus::ALLSPARK_RUNTIME_ERROR;
      const_cast<AsTensor*>(static_cast<const AsTensor*>(weights_[0]))->Free();
    }
    return AsStatus::ALLSPARK_SUCCESS;
  }

  AsStatus Reshape(RuntimeContext*) override {
    AsTensor* x = tensor_map_->at(in_names_[0]).get();
    Shape yshape = x->GetShape();
    if (yshape.empty() || (int)yshape.back() != k_) return AsStatus::ALLSPARK_PARAM_ERROR;
    m_ = (int)(x->Count() / k_);
    yshape.back() = n_;
    AsTensor* y = tensor_map_->at(out_names_[0]).get();
    y->SetDataType(x->GetDataType());
    AS_CHECK_STATUS(y->SetShape(std::move(yshape)));
    const int64_t ws = (int64_t)dihip_gemm_lowp_workspace_bytes(WBITS, std::max(m_, 1), n_, k_, group_size_);
    AsTensor* wsp = tensor_map_->at("workspace").get();  // shared scratch, grows only (model.cpp:241-243)
    if (wsp->GetSizeInByte() < (size_t)ws) AS_CHECK_STATUS(wsp->SetShape(Shape{ws}));
    return AsStatus::ALLSPARK_SUCCESS;
  }

  AsStatus Forward(RuntimeContext*) override {
    AsTensor* x = tensor_map_->at(in_names_[0]).get();
    AsTensor* y = tensor_map_->at(out_names_[0]).get();
    AsTensor* wsp = tensor_map_->at("workspace").get();
    const void* bias = weights_.size() == 4 ? weights_[3]->GetDataPtr() : nullptr;
    // fused binary ADD (the residual): under tensor parallelism only rank 0 applies it -- the partial results of the
    // ranks are summed by the AllReduce that follows, and the residual must enter that sum once
    // (GemmOpBase::Reshape, csrc/core/operator/general/gemm/gemm_op.cpp:133-137: binary_type_ = UNDEFINED on rank != 0)
    const bool add_residual = in_names_.size() > 1 && (ctx_->GetNranks() <= 1 || ctx_->GetRank() == 0);
    const void* residual = add_residual ? tensor_map_->at(in_names_[1])->GetDataPtr() : nullptr;
    if (x->GetDataType() != ftype_) return AsStatus::ALLSPARK_PARAM_ERROR;
    hipStream_t s = static_cast<const HIPContext*>(ctx_)->GetStream();
    auto fn = WBITS == 8 ? dihip_gemm_a16w8 : dihip_gemm_a16w4;
    return FromDihip(fn(s, x->GetDataPtr(), packed_w_->GetDataPtr(), packed_sz_->GetDataPtr(), bias, residual, y->GetDataPtr(), m_,
                        n_, k_, group_size_, (int)activation_, alpha_, wsp->GetDataPtr(), wsp->GetSizeInByte(),
                        sync_->GetDataPtr(), DihipDtype(ftype_)));
  }

 private:
  int m_ = 0, n_ = 0, k_ = 0, group_size_ = -1;
  float alpha_ = 1.0f;
  UnaryType activation_ = UNARYTYPE_UNDEFINED;
  DataType ftype_ = BFLOAT16;
  std::unique_ptr<AsTensor> packed_w_, packed_sz_, sync_;
};

using GemmA16W8HIP = GemmLowpHIP<8>;
using GemmA16W4HIP = GemmLowpHIP<4>;
REGISTER_OP(GemmA16W8, HIP, GemmA16W8HIP)
REGISTER_OP(GemmA16W4, HIP, GemmA16W4HIP)

}  // namespace allspark
