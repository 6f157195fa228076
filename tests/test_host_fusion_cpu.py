"""The fusion pass of the HIP operator layer (dash-infer_amd/host/fusion_pass.cpp) on CPU: it rewrites the reference's Qwen2
operator list (tests/ref_graph.py: python/pyhie/allspark/model/qwen_v15.py:187-388) into the fused decode-step operators, follows
tensors by name, threads the handed-on norm from layer to layer, keeps the AllReduce behind a row-parallel projection, and leaves
a list it does not fully understand UNCHANGED with the first misfit named."""
import pytest

from tests import ref_graph


@pytest.fixture()
def model(pkg):
    from dash_infer_amd import hostapi
    m = hostapi.Model(None, 4, 2, 128, 16)
    yield m
    m.close()


def test_two_layer_graph_is_fused_five_operators_per_layer(model):
    g = ref_graph.qwen2_graph(2, 4, 128, 1e-6, 4, 2, 1000000.0)
    ref_graph.add_graph(model, g)
    r = model.graph_fuse_dry()
    assert r["fused"] and r["device_resident"] and r["layers"] == 2, r["why"]
    per_layer = ["DihipNormGemm", "DihipRopeSpanAttn", "DihipGemmAddTo", "DihipNormSwiGLU", "DihipGemmAddTo"]
    assert r["types"] == ["DihipEmbedding"] + per_layer * 2 + ["DihipLMHead", "DihipGreedy"]
    assert r["ops"] == f"{len(g)}->13"
    w = r["wiring"].split("|")
    # layer 0 reads the hidden rows only; its down projection hands the next layer's norm on; layer 1 takes it; the last does not
    assert w[1].startswith("DihipNormGemm(embedding.out)->(decoder.layer.0.attention.self.out)[decoder.layer.0.attention.layernorm.gamma,")
    assert "decoder.layer.0.attention.self.bias]" in w[1]
    assert w[2] == "DihipRopeSpanAttn(decoder.layer.0.attention.self.out)->(decoder.layer.0.attention.out)[]"
    assert w[3].startswith("DihipGemmAddTo(decoder.layer.0.attention.out,embedding.out)->(decoder.layer.0.attention_add.out,decoder.layer.0.attention.output.dense.dihip_xnorm)")
    assert w[3].endswith("decoder.layer.0.ffn.layernorm.gamma]")
    assert w[4].startswith("DihipNormSwiGLU(decoder.layer.0.attention_add.out,decoder.layer.0.attention.output.dense.dihip_xnorm)->(decoder.layer.0.ffn.mul.out)")
    assert w[5].startswith("DihipGemmAddTo(decoder.layer.0.ffn.mul.out,decoder.layer.0.attention_add.out)->(decoder.layer.0.final_add.out,decoder.layer.0.ffn.output.dense.dihip_xnorm)")
    assert w[5].endswith("decoder.layer.1.attention.layernorm.gamma]")
    assert w[6].startswith("DihipNormGemm(decoder.layer.0.final_add.out,decoder.layer.0.ffn.output.dense.dihip_xnorm)->")
    assert w[10].startswith("DihipGemmAddTo(decoder.layer.1.ffn.mul.out,decoder.layer.1.attention_add.out)->(decoder.layer.1.final_add.out)[")
    assert w[10].endswith("decoder.layer.1.ffn.output.dense.weight.zero_point]")
    assert w[11] == "DihipLMHead(decoder.layer.1.final_add.out)->(logits)[final.layernorm.gamma,lm_head.weight]"
    assert w[12] == "DihipGreedy(logits)->(generated_ids)[]"


def test_allreduce_stays_behind_the_row_parallel_projections(pkg):
    from dash_infer_amd import hostapi
    m = hostapi.Model(None, 4, 2, 128, 16, rank=1, nranks=2)
    ref_graph.add_graph(m, ref_graph.qwen2_graph(1, 8, -1, 1e-6, 4, 2, 1e6, tp_allreduce=True, tp_lm_head=True))
    r = m.graph_fuse_dry()
    # tensor parallel: the layers are fused (AllReduce kept behind the row-parallel projections), the K-split lm_head + its
    # AllReduce stay the reference's own operators behind a DihipFinalNorm; since round 6 the sampling operator behind them is
    # DihipGreedy (FT logits cast to f32, device-resident length counters): the step state stays on the device and the step replays as a graph
    assert r["fused"] and r["device_resident"], r["why"]
    assert r["types"] == ["DihipEmbedding", "DihipNormGemm", "DihipRopeSpanAttn", "DihipGemmAddTo", "AllReduce", "DihipNormSwiGLU",
                          "DihipGemmAddTo", "AllReduce", "DihipFinalNorm", "GetLastLine", "Gemm", "AllReduce", "DihipGreedy"]
    w = r["wiring"].split("|")
    assert w[8] == "DihipFinalNorm(decoder.layer.0.final_add.out)->(last_hidden_state)[final.layernorm.gamma]"
    assert "single-rank" in r["why"] or "splitk" in r["why"] or "lm_head" in r["why"]
    m.close()
    m = hostapi.Model(None, 4, 2, 128, 16)   # the same list on one rank: the AllReduce operators are kept (they copy)
    ref_graph.add_graph(m, ref_graph.qwen2_graph(1, 8, -1, 1e-6, 4, 2, 1e6, tp_allreduce=True))
    r = m.graph_fuse_dry()
    assert r["fused"] and r["device_resident"], r["why"]
    assert r["types"] == ["DihipEmbedding", "DihipNormGemm", "DihipRopeSpanAttn", "DihipGemmAddTo", "AllReduce", "DihipNormSwiGLU",
                          "DihipGemmAddTo", "AllReduce", "DihipLMHead", "DihipGreedy"]
    w = r["wiring"].split("|")
    assert w[4] == "AllReduce(decoder.layer.0.attention_add.out)->(decoder.layer.0.attention_add.out)[]"   # in place on the f32 rows
    m.close()


@pytest.mark.parametrize("mutate,needle", [
    (lambda g: g.__setitem__(3, ("Rotary", g[3][1], g[3][2], g[3][3], [], g[3][5] + ";rotary_type=i:2")), "Rotary variant"),
    (lambda g: g.__setitem__(5, (g[5][0], g[5][1], g[5][2], g[5][3], g[5][4] + ["x.bias"], g[5][5])), "unexpected weights"),
    (lambda g: g.__setitem__(8, (g[8][0], g[8][1], g[8][2], g[8][3], g[8][4], "GroupSize=i:128;alpha=f:1.0")), "unexpected activation"),
    (lambda g: g.__setitem__(6, ("Binary", g[6][1], [g[6][2][0], "something.else"], g[6][3], [], g[6][5])), "does not combine"),
    (lambda g: g.insert(4, ("Unary", "extra", ["decoder.layer.0.rotary.out"], ["decoder.layer.0.rotary.out"], [], "unary_type=i:4")), "DecOptMQA"),
    (lambda g: g.pop(0), "EmbeddingT5"),
])
def test_a_list_that_does_not_fit_is_left_unchanged(model, mutate, needle):
    g = ref_graph.qwen2_graph(1, 4, 128, 1e-6, 4, 2, 1e6)
    mutate(g)
    ref_graph.add_graph(model, g)
    r = model.graph_fuse_dry()
    assert not r["fused"]
    assert needle in r["why"], r["why"]
    assert r["types"] == [t[0] for t in g]          # unchanged, operator by operator


def test_a_tail_with_an_unknown_operator_is_refused(model):
    g = ref_graph.qwen2_graph(1, 4, 128, 1e-6, 4, 2, 1e6)
    g.insert(len(g) - 1, ("Unary", "logit_scale", ["logits"], ["logits"], [], "unary_type=i:4"))
    ref_graph.add_graph(model, g)
    r = model.graph_fuse_dry()
    assert not r["fused"] and "Unary" in r["why"]
    assert r["types"] == [t[0] for t in g]


def test_split_k_lm_head_on_one_rank_keeps_the_reference_tail(model):
    g = ref_graph.qwen2_graph(1, 4, 128, 1e-6, 4, 2, 1e6, tp_lm_head=True)
    ref_graph.add_graph(model, g)
    r = model.graph_fuse_dry()
    assert r["fused"] and r["device_resident"]
    assert r["types"][-5:] == ["DihipFinalNorm", "GetLastLine", "Gemm", "AllReduce", "DihipGreedy"]


def test_mixture_of_experts_layer_becomes_one_block_operator(pkg):
    """qwen_v20_moe.py:318-391: the twelve feed-forward operators of a MoE layer -> DihipMoeBlock; the reference's two
    all-reduces (MOE rows, CalcExpert rows) become ONE all-reduce of the f32 hidden rows behind the block"""
    from dash_infer_amd import hostapi
    m = hostapi.Model(None, 4, 2, 128, 16)
    g = ref_graph.qwen2_graph(2, 8, 128, 1e-6, 4, 2, 1e6, moe=(8, 2))
    assert len(g) == 1 + 2 * (6 + 10) + 4
    ref_graph.add_graph(m, g)
    r = m.graph_fuse_dry()
    assert r["fused"] and r["device_resident"] and r["layers"] == 2, r["why"]
    per_layer = ["DihipNormGemm", "DihipRopeSpanAttn", "DihipGemmAddTo", "DihipMoeBlock"]
    assert r["types"] == ["DihipEmbedding"] + per_layer * 2 + ["DihipLMHead", "DihipGreedy"]
    w = r["wiring"].split("|")
    p = "decoder.layer.0."
    assert w[3].startswith(f"DihipGemmAddTo({p}attention.out,embedding.out)->({p}attention_add.out)[")    # no norm handed on
    assert w[4].startswith(f"DihipMoeBlock({p}attention_add.out)->({p}final_add.out)[{p}ffn.layernorm.gamma,{p}mlp.gate.weight,{p}mlp.experts.gate_up_proj.weight,")
    assert w[4].endswith(f"{p}shared_expert.down_proj.weight.zero_point,{p}shared_expert_gate.weight]") and w[4].count(",") == 14
    assert w[5].startswith(f"DihipNormGemm({p}final_add.out)->")                                           # the next layer norms itself
    m.close()
    # expert parallelism on two ranks: AllReduce after the attention projection and ONE after the block
    m = hostapi.Model(None, 4, 2, 128, 16, rank=1, nranks=2)
    ref_graph.add_graph(m, ref_graph.qwen2_graph(1, 8, -1, 1e-6, 4, 2, 1e6, tp_allreduce=True, tp_lm_head=True, moe=(8, 2, True)))
    r = m.graph_fuse_dry()
    assert r["fused"] and r["device_resident"], r["why"]
    assert r["types"][:7] == ["DihipEmbedding", "DihipNormGemm", "DihipRopeSpanAttn", "DihipGemmAddTo", "AllReduce", "DihipMoeBlock", "AllReduce"]
    w = r["wiring"].split("|")
    assert w[6] == "AllReduce(decoder.layer.0.final_add.out)->(decoder.layer.0.final_add.out)[]"
    m.close()
    # a block the pass does not know (CalcExpert on other tensors) leaves the whole list alone
    m = hostapi.Model(None, 4, 2, 128, 16)
    g = ref_graph.qwen2_graph(1, 8, 128, 1e-6, 4, 2, 1e6, moe=(8, 2))
    k = next(i for i, op in enumerate(g) if op[0] == "CalcExpert")
    g[k] = (g[k][0], g[k][1], [g[k][2][1], g[k][2][0]], g[k][3], g[k][4], g[k][5])
    ref_graph.add_graph(m, g)
    r = m.graph_fuse_dry()
    assert not r["fused"] and "CalcExpert" in r["why"], r["why"]
    m.close()


def test_the_converters_own_arities_are_accepted(pkg):
    """decoder graph + gen_graph as the reference's converter writes them (ref_graph.as_exported): Rotary with the position
    mask, the attention with GenerateOp's beam index appended, GenerateOp with two inputs / three outputs, UpdateId behind it --
    same fused list; UpdateId is the model runner's (stop checks need the token on the host)"""
    from dash_infer_amd import hostapi
    m = hostapi.Model(None, 4, 2, 128, 16)
    plain = ref_graph.qwen2_graph(2, 4, 128, 1e-6, 4, 2, 1e6)
    g = ref_graph.as_exported(plain)
    assert len(g) == len(plain) + 1 and g[-1][0] == "UpdateId" and len(g[-2][3]) == 3 and len(g[4][2]) == 3
    ref_graph.add_graph(m, g)
    r = m.graph_fuse_dry()
    assert r["fused"] and r["device_resident"] and r["layers"] == 2, r["why"]
    per_layer = ["DihipNormGemm", "DihipRopeSpanAttn", "DihipGemmAddTo", "DihipNormSwiGLU", "DihipGemmAddTo"]
    assert r["types"] == ["DihipEmbedding"] + per_layer * 2 + ["DihipLMHead", "DihipGreedy"]
    assert "UpdateId" in r["why"]
    w = r["wiring"].split("|")
    assert w[2] == "DihipRopeSpanAttn(decoder.layer.0.attention.self.out)->(decoder.layer.0.attention.out)[]"
    assert w[12] == "DihipGreedy(logits)->(generated_ids,generate.next_beam_idx,generate.hyps)[]"
    m.close()
    # the same with a tensor-parallel tail and with MoE layers
    m = hostapi.Model(None, 4, 2, 128, 16, rank=0, nranks=2)
    ref_graph.add_graph(m, ref_graph.as_exported(ref_graph.qwen2_graph(1, 8, -1, 1e-6, 4, 2, 1e6, tp_allreduce=True, tp_lm_head=True, moe=(8, 2, True))))
    r = m.graph_fuse_dry()
    assert r["fused"] and r["device_resident"], r["why"]
    assert r["types"][-5:] == ["DihipFinalNorm", "GetLastLine", "Gemm", "AllReduce", "DihipGreedy"] and "DihipMoeBlock" in r["types"]
    m.close()


def test_the_decoder_graph_alone_fuses_up_to_the_logits(pkg):
    """AsModel keeps GenerateOp / UpdateId in gen_graph: a pass run per graph sees a list that ends in the lm_head Gemm"""
    from dash_infer_amd import hostapi
    m = hostapi.Model(None, 4, 2, 128, 16)
    g = [op for op in ref_graph.qwen2_graph(1, 4, 128, 1e-6, 4, 2, 1e6) if op[0] != "GenerateOp"]
    ref_graph.add_graph(m, g)
    r = m.graph_fuse_dry()
    assert r["fused"] and not r["device_resident"] and r["types"][-1] == "DihipLMHead" and "DihipGreedy" not in r["types"], r["why"]
    assert "GenerateOp" in r["why"]
    m.close()
