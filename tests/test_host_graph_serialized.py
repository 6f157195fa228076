"""A SERIALIZED allspark `TransformerProto` whose graphs were built by the reference's own converter (tests/golden/make_graph_golden.py
runs `Qwen_v15._build_graph` + `quantize_op` of /root/reference over the message classes of dash-infer_amd/graph_proto.py and commits
the bytes) goes into the C++ operator layer through host/graph_wire.h -- the ingest AsModel does with the generated protobuf
classes (csrc/core/model/model.cpp:265-287) -- and the fusion pass accepts it AS EXPORTED: pre-processing graphs apart, in-place
RichEmbedding, 8-byte integer attributes (torch.tensor(int).numpy().tobytes(), model_base.py:64-66), the converter's weight names.
CPU only (the fusion pass without creating operators); the GPU run of the same bytes is tests/test_gpu_host_runner.py."""
import os
import struct

import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _bytes(name):
    return open(os.path.join(GOLDEN, name), "rb").read()


def test_python_reader_and_cpp_wire_reader_see_the_same_operator_lists(pkg):
    from dash_infer_amd import graph_proto as gp, hostapi
    data = _bytes("qwen2_a16w4_g128.asgraph.pb")
    lists = dict(gp.ops_of(data))
    assert list(lists) == ["pre_graph", "decoder", "gen_graph", "post_graph"]
    assert [o.op_type for o in lists["pre_graph"]] == ["PreProcessId", "UpdateId", "TransMask"]
    dec = [gp.op_as_tuple(o) for o in lists["decoder"]]
    assert [t[0] for t in dec[:8]] == ["EmbeddingT5", "RichEmbedding", "LayerNormNoBeta", "GemmA16W4", "Rotary", "DecOptMQA", "GemmA16W4", "Binary"]
    qkv = dec[3]
    assert qkv[4] == ["decoder.layer.0.attention.self.weight", "decoder.layer.0.attention.self.weight.scale",
                      "decoder.layer.0.attention.self.weight.zero_point", "decoder.layer.0.attention.self.bias"]
    assert struct.unpack("<i", qkv[5]["GroupSize"])[0] == 128 and qkv[5]["with_bias"] == b"\x01"
    assert struct.unpack("<q", dec[5][5]["num_heads"])[0] == 28          # Python ints travel as int64: the C++ side reads the low 4 bytes
    # the same bytes through the C++ wire reader + fusion pass (no operator is created: works without a GPU)
    m = hostapi.Model(None, 28, 4, 128, 128)
    try:
        m.graph_add_serialized(data)                                       # decoder + gen_graph, as AsModel runs them per step
        r = m.graph_fuse_dry()
    finally:
        m.close()
    assert r["fused"] and r["device_resident"] and r["layers"] == 2, r["why"]
    per_layer = ["DihipNormGemm", "DihipRopeSpanAttn", "DihipGemmAddTo", "DihipNormSwiGLU", "DihipGemmAddTo"]
    assert r["types"] == ["DihipEmbedding"] + per_layer * 2 + ["DihipLMHead", "DihipGreedy"]
    assert r["ops"] == "31->13"                                           # 29 decoder operators + GenerateOp + UpdateId
    w = r["wiring"].split("|")
    assert w[0] == "DihipEmbedding(dec_ids)->(embedding.out)[embedding.word_embeddings]"
    assert w[1].startswith("DihipNormGemm(embedding.out)->(decoder.layer.0.attention.self.out)[decoder.layer.0.attention.layernorm.gamma,"
                           "decoder.layer.0.attention.self.weight,decoder.layer.0.attention.self.weight.scale,")
    assert w[11] == "DihipLMHead(decoder.layer.1.final_add.out)->(logits)[final.layernorm.gamma,lm_head.weight]"
    assert w[12].startswith("DihipGreedy(logits)->(dec_ids")             # the converter's in-place id tensor (the runner renames the output)


def test_int8_per_channel_export_fuses_too_and_garbage_is_refused(pkg):
    from dash_infer_amd import hostapi
    m = hostapi.Model(None, 28, 4, 128, 128)
    try:
        m.graph_add_serialized(_bytes("qwen2_a16w8_perc.asgraph.pb"), graphs=["decoder", "gen_graph"])
        r = m.graph_fuse_dry()
        assert r["fused"] and r["layers"] == 2, r["why"]
        with pytest.raises(Exception):
            m.graph_add_serialized(b"\xff\xff\xff not a protobuf")
        with pytest.raises(Exception):
            m.graph_add_serialized(_bytes("qwen2_a16w8_perc.asgraph.pb"), graphs=["no_such_graph"])
    finally:
        m.close()


def test_the_hand_written_list_and_the_export_agree(pkg):
    """dash-infer_amd/ref_graph.py (what bench.py feeds the operator layer) against the converter's own decoder graph: same operator
    types in the same order and the same weight names, the export's extra operators (RichEmbedding) and arities apart"""
    from dash_infer_amd import graph_proto as gp, ref_graph
    exported = [gp.op_as_tuple(o) for o in dict(gp.ops_of(_bytes("qwen2_a16w4_g128.asgraph.pb")))["decoder"]]
    mine = ref_graph.qwen2_graph(2, 4, 128, 1e-6, 28, 4, 1000000.0)
    exp = [t for t in exported if t[0] != "RichEmbedding"]
    assert [t[0] for t in exp] == [t[0] for t in mine[:len(exp)]]
    for a, b in zip(exp, mine):
        assert a[1] == b[1] and a[4] == list(b[4]), (a[1], a[4], b[4])
