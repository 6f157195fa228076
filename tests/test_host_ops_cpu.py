"""CPU checks of the C++ operator layer: the library loads, exports every symbol its header
declares, the OpFactory holds exactly the hot-path op types for DeviceType::HIP and rejects
anything else the way the reference's factory does (operator.cpp:379-386)."""
import pytest


def test_ops_library_exports_header_symbols(pkg):
    from dash_infer_amd import hostapi
    l = hostapi.lib()
    missing = [s for s in hostapi.header_symbols() if not hasattr(l, s)]
    assert not missing, missing


def test_op_registry(pkg):
    from dash_infer_amd import hostapi
    ops = hostapi.lib().dihost_registered_ops().decode().split(",")
    # every op type of the Qwen2 generation graph (qwen_v15.py:187-388, model_base.py:690-703) resolves for DeviceType::HIP,
    # the graph-head / id-processing operators included (RichEmbedding -- multimodal inputs -- is not: INTEGRATION.md section 3)
    reference_types = ["GemmA16W8", "GemmA16W4", "DecOptMHA", "DecOptMQA", "AllReduce", "AllGather", "MOEA16W8", "CalcExpert",
                       "Gemm", "Rotary", "LayerNormNoBeta", "Binary", "Unary", "UnaryGLU", "EmbeddingT5", "GetLastLine", "GenerateOp",
                       "TransMask", "PreProcessId", "UpdateId", "PostProcessId"]   # graph head / tail (host/id_ops_hip.cpp)
    # + the fused decode-step operators the fusion pass rewrites that graph into (host/fused_ops_hip.cpp)
    fused_types = ["DihipEmbedding", "DihipNormGemm", "DihipRopeSpanAttn", "DihipGemmAddTo", "DihipNormSwiGLU", "DihipLMHead", "DihipGreedy",
                   "DihipFinalNorm", "DihipMoeBlock"]
    assert sorted(t for t in ops if not t.startswith("Dihip")) == sorted(reference_types)
    assert sorted(t for t in ops if t.startswith("Dihip")) == sorted(fused_types)


def test_unknown_op_type_is_rejected(pkg):
    from dash_infer_amd import hostapi
    m = hostapi.Model(None, 4, 2, 128, 16)
    with pytest.raises(hostapi.HostError) as e:
        m.create_op("RichEmbedding", "rich_embedding", ["x"], ["y"])
    assert "Unsupported op type." in str(e.value) and e.value.code == 2
    # a span op whose name carries no layer index fails Init with PARAM_ERROR (span_attn_op.cpp:182-186)
    with pytest.raises(hostapi.HostError) as e:
        m.create_op("DecOptMQA", "attention", ["qkv"], ["out"])
    assert e.value.code == 2
    m.close()
