// fusion_pass.h -- rewrites the reference's Qwen2 operator list (python/pyhie/allspark/model/qwen_v15.py:187-388 in its
// weight-only-quantised form, model_base.py:690-703 tail) into the fused decode-step operators of host/fused_ops_hip.cpp.
// Runs where AsModel builds its operator list from the graph proto (csrc/core/model/model.cpp:265-287), before OpFactory look-up.
#pragma once
#include <string>
#include <vector>

#include "as_types.h"

namespace allspark {

struct FusionReport {
  bool fused = false;   // false: the list is returned unchanged, `why` names the first operator that did not fit
  // true: the tail is fused as well (DihipLMHead + DihipGreedy), so a decoder step keeps its whole state on the device (sequence
  // lengths advanced by the sampling launch) and may be captured as a hipGraph.  false with fused == true: the layers are fused
  // but the reference's own tail operators run behind a DihipFinalNorm (tensor-parallel lm_head: Gemm with splitk + AllReduce);
  // lengths are then staged from the host per step like the unfused operators do.
  bool device_resident = false;
  int layers = 0;
  int ops_before = 0, ops_after = 0;
  std::string why;
};

// Pattern table (one decoder layer; tensors are followed by NAME, so operator names are free):
//   LayerNormNoBeta(h) , GemmA16W8|W4(.; bias)                      -> DihipNormGemm(h [, xnorm of the previous layer])
//   Rotary(.) , DecOptMQA|DecOptMHA(.)                               -> DihipRopeSpanAttn
//   GemmA16Wx(.) , [AllReduce] , Binary ADD(., h)                    -> DihipGemmAddTo(+ gamma of the next LayerNormNoBeta) [, AllReduce]
//   LayerNormNoBeta(h') , GemmA16Wx(.; SILU) , GemmA16Wx(.) , Binary MUL -> DihipNormSwiGLU(h' [, xnorm])
//   GemmA16Wx(.) , [AllReduce] , Binary ADD(., h')                   -> DihipGemmAddTo(+ gamma of the next layer's first norm) [, AllReduce]
// head / tail: EmbeddingT5 -> DihipEmbedding ;  LayerNormNoBeta , GetLastLine , Gemm(lm_head) -> DihipLMHead ; GenerateOp -> DihipGreedy
// any other tail made of GetLastLine / Gemm / AllReduce / GenerateOp (the tensor-parallel lm_head of model_base.py:690-703: Gemm with
// splitk + AllReduce) keeps those operators behind  LayerNormNoBeta -> DihipFinalNorm  (f32 hidden rows -> FT rows).
// Anything else (another rotary variant, a bias on o / down, alpha != 1, ...) leaves the WHOLE list unfused.
std::vector<OperatorProto> FuseDecoderGraph(const std::vector<OperatorProto>& graph, const DeviceContext& ctx, FusionReport* report);

}  // namespace allspark
