"""The reference's Qwen2 operator list (python/pyhie/allspark/model/qwen_v15.py:187-388 in its weight-only-quantised form --
dynamic quantisation switches the Gemm's fused binary ADD off, qwen_v15.py:175-178 -- and the tail of model_base.py:690-703) as
plain tuples (op_type, op_name, inputs, outputs, weights, attrs): what bench.py's host runner and the host-layer tests feed to the
C++ operator layer (hostapi.Model.graph_build).  Pure Python: the CPU tests of the fusion pass use it without a GPU."""


def qwen2_graph(n_layers, wbits, group, eps, n_heads, n_kv, rope_theta, tp_allreduce=False, tp_lm_head=False, moe=None):
    """moe = (num_experts, top_k[, use_ep]): the feed-forward half of every layer is the mixture-of-experts block of
    python/pyhie/allspark/model/qwen_v20_moe.py:318-391 (MOE in its weight-only form MOEA16W8)."""
    gemm = "GemmA16W4" if wbits == 4 else "GemmA16W8"
    gattr = f"GroupSize=i:{group}" if group and group > 0 else ""

    def lowp(name, inp, out, act=0, bias=False):
        w = [name + ".weight", name + ".weight.scale", name + ".weight.zero_point"] + ([name + ".bias"] if bias else [])
        attrs = ";".join(a for a in (gattr, f"activation=i:{act}" if act else "", "alpha=f:1.0") if a)
        return (gemm, name, [inp], [out], w, attrs)

    g = [("EmbeddingT5", "embedding", ["input_ids"], ["embedding.out"], ["embedding.word_embeddings"], "token_embedding=b:0")]
    prev = "embedding.out"
    for li in range(n_layers):
        p = f"decoder.layer.{li}."
        g.append(("LayerNormNoBeta", p + "attention.layernorm", [prev], [p + "attention.layernorm.out"], [p + "attention.layernorm.gamma"], f"eps=f:{eps}"))
        g.append(lowp(p + "attention.self", p + "attention.layernorm.out", p + "attention.self.out", bias=True))
        g.append(("Rotary", p + "rotary", [p + "attention.self.out"], [p + "rotary.out"], [],
                  f"num_heads=i:{n_heads};multi_query_group_num=i:{n_kv};rotary_base=f:{rope_theta}"))
        g.append(("DecOptMQA", p + "attention", [p + "rotary.out"], [p + "attention.out"], [], ""))
        g.append(lowp(p + "attention.output.dense", p + "attention.out", p + "attention.output.dense.out"))
        o_out = p + "attention.output.dense.out"
        if tp_allreduce:
            g.append(("AllReduce", p + "attention.all_reduce", [o_out], [o_out], [], ""))
        g.append(("Binary", p + "attention_add", [o_out, prev], [p + "attention_add.out"], [], "binary_type=i:1"))
        g.append(("LayerNormNoBeta", p + "ffn.layernorm", [p + "attention_add.out"], [p + "ffn.layernorm.out"], [p + "ffn.layernorm.gamma"], f"eps=f:{eps}"))
        if moe is not None:
            xn, ex = p + "ffn.layernorm.out", p + "mlp.experts"
            g.append(("Gemm", p + "mlp.gate", [xn], [p + "mlp.gate.out"], [p + "mlp.gate.weight"], "with_bias=b:0"))
            mattr = f"num_experts=i:{moe[0]};num_experts_per_tok=i:{moe[1]}" + (";use_ep=b:1" if len(moe) > 2 and moe[2] else "") + (";" + gattr if gattr else "")
            g.append(("MOEA16W8", ex, [xn, p + "mlp.gate.out"], [ex + ".out"],
                      [ex + ".gate_up_proj.weight", ex + ".gate_up_proj.weight.scale", ex + ".gate_up_proj.weight.zero_point",
                       ex + ".down_proj.weight", ex + ".down_proj.weight.scale", ex + ".down_proj.weight.zero_point"], mattr))
            if tp_allreduce:
                g.append(("AllReduce", p + "attention.all_reduce_moe", [ex + ".out"], [ex + ".out"], [], ""))
            g.append(lowp(p + "shared_expert.gate_up_proj", xn, p + "shared_expert.gate_up_proj.out"))
            g.append(("UnaryGLU", p + "shared_expert_act_mul", [p + "shared_expert.gate_up_proj.out"], [p + "shared_expert_act_mul.out"], [], "unary_type=i:5"))
            g.append(lowp(p + "shared_expert.down_proj", p + "shared_expert_act_mul.out", p + "shared_expert.down_proj.out"))
            g.append(("Gemm", p + "shared_expert_gate", [xn], [p + "shared_expert_gate.out"], [p + "shared_expert_gate.weight"], "with_bias=b:0;activation=i:6"))
            g.append(("CalcExpert", p + "shared_calc_expert", [p + "shared_expert.down_proj.out", p + "shared_expert_gate.out"], [p + "shared_calc_expert.out"], [],
                      "num_experts=i:1"))
            if tp_allreduce:
                g.append(("AllReduce", p + "all_reduce_shared_expert", [p + "shared_calc_expert.out"], [p + "shared_calc_expert.out"], [], ""))
            g.append(("Binary", p + "expert_add", [ex + ".out", p + "shared_calc_expert.out"], [p + "expert_add.out"], [], "binary_type=i:1"))
            g.append(("Binary", p + "final_add", [p + "expert_add.out", p + "attention_add.out"], [p + "final_add.out"], [], "binary_type=i:1"))
            prev = p + "final_add.out"
            continue
        g.append(lowp(p + "ffn.intermediate.dense", p + "ffn.layernorm.out", p + "ffn.intermediate.dense.out", act=5))
        g.append(lowp(p + "ffn.linear.dense", p + "ffn.layernorm.out", p + "ffn.linear.dense.out"))
        g.append(("Binary", p + "ffn.mul", [p + "ffn.intermediate.dense.out", p + "ffn.linear.dense.out"], [p + "ffn.mul.out"], [], "binary_type=i:2"))
        g.append(lowp(p + "ffn.output.dense", p + "ffn.mul.out", p + "ffn.output.dense.out"))
        d_out = p + "ffn.output.dense.out"
        if tp_allreduce:
            g.append(("AllReduce", p + "ffn.all_reduce", [d_out], [d_out], [], ""))
        g.append(("Binary", p + "final_add", [d_out, p + "attention_add.out"], [p + "final_add.out"], [], "binary_type=i:1"))
        prev = p + "final_add.out"
    g.append(("LayerNormNoBeta", "final.layernorm", [prev], ["last_hidden_state"], ["final.layernorm.gamma"], f"eps=f:{eps}"))
    g.append(("GetLastLine", "get_last_line", ["last_hidden_state"], ["get_last_line.out"], [], ""))
    if tp_lm_head:   # model_base.py:690-703: K-split lm_head (Gemm with splitk, lm_head.weight HSPLIT) + AllReduce of the logits
        g.append(("Gemm", "lm_head", ["get_last_line.out"], ["lm_head.out"], ["lm_head.weight"], "with_bias=b:0;splitk=b:1"))
        g.append(("AllReduce", "all_reduce_lmhead", ["lm_head.out"], ["logits"], [], ""))
    else:
        g.append(("Gemm", "lm_head", ["get_last_line.out"], ["logits"], ["lm_head.weight"], "with_bias=b:0"))
    g.append(("GenerateOp", "generate", ["logits"], ["generated_ids"], [], "top_k=i:1"))
    return g


def as_exported(graph):
    """The same list with the arities the reference's converter actually writes (qwen_v15.py:408-452, model_base.py GenerateOp):
    Rotary also takes the position mask of TransMask (which lives in pre_graph), the attention operator has GenerateOp's beam
    index appended to its inputs, GenerateOp takes the original ids as a second input and declares three outputs, and gen_graph
    ends in UpdateId -- decoder graph and gen_graph back to back, as AsModel runs them."""
    out = []
    for t, name, inputs, outputs, weights, attrs in graph:
        inputs, outputs = list(inputs), list(outputs)
        if t == "Rotary":
            inputs.append("transmask.out1")
        elif t in ("DecOptMQA", "DecOptMHA"):   # [rotary out, attention mask, + the beam index appended at qwen_v15.py:445-447]
            inputs += ["transmask.out", "generate.next_beam_idx"]
            attrs = ";".join(a for a in (attrs, "size_per_head=i:128", "multigpu=i:1") if a)
        elif t == "GenerateOp":
            inputs.append("preprocess_id.out1")
            outputs += ["generate.next_beam_idx", "generate.hyps"]
        out.append((t, name, inputs, outputs, weights, attrs))
    out.append(("UpdateId", "update_id", ["preprocess_id.out", "generate.next_beam_idx"], ["update_id.out"], [], ""))
    return out


def register_weights(m, model, ft="bf16"):
    """The product model's unpacked quantised weights (decoder.build_random_model(keep_fp=True)) under the reference's names."""
    fp = model.fp
    qdt = "u8" if model.quant.wbits == 4 else "i8"

    def lowp(name, key, li):
        q, s, z = fp[li][key]
        m.set_weight(name + ".weight", q, qdt)
        m.set_weight(name + ".weight.scale", s, ft)
        m.set_weight(name + ".weight.zero_point", z, ft)

    m.set_weight("embedding.word_embeddings", fp["embed"], ft)
    for li in range(len(model.layers)):
        p = f"decoder.layer.{li}."
        m.set_weight(p + "attention.layernorm.gamma", fp[li]["ln1"], ft)
        m.set_weight(p + "ffn.layernorm.gamma", fp[li]["ln2"], ft)
        lowp(p + "attention.self", "qkv", li)
        m.set_weight(p + "attention.self.bias", fp[li]["qkv_bias"], ft)
        lowp(p + "attention.output.dense", "o", li)
        if "moe" in fp[li]:
            import torch
            mo, ex = fp[li]["moe"], p + "mlp.experts"
            m.set_weight(p + "mlp.gate.weight", mo["router"], ft)
            m.set_weight(p + "shared_expert_gate.weight", mo["shared_gate_w"], ft)
            cat = lambda a, b: [torch.cat([x, y], dim=1).contiguous() for x, y in zip(a, b)]      # columns [gate | up] (unary.cu:122-132)
            gq, gs, gz = cat(fp[li]["gate"], fp[li]["up"])
            for name, t, dt in ((".weight", gq, qdt), (".weight.scale", gs, ft), (".weight.zero_point", gz, ft)):
                m.set_weight(p + "shared_expert.gate_up_proj" + name, t, dt)
            lowp(p + "shared_expert.down_proj", "down", li)
            gu = [cat(g_, u_) for g_, u_ in zip(mo["experts_gate"], mo["experts_up"])]
            for j, suffix in enumerate((".weight", ".weight.scale", ".weight.zero_point")):
                m.set_weight(ex + ".gate_up_proj" + suffix, torch.stack([e[j] for e in gu]).contiguous(), qdt if j == 0 else ft)
                m.set_weight(ex + ".down_proj" + suffix, torch.stack([e[j] for e in mo["experts_down"]]).contiguous(), qdt if j == 0 else ft)
            continue
        lowp(p + "ffn.intermediate.dense", "gate", li)
        lowp(p + "ffn.linear.dense", "up", li)
        lowp(p + "ffn.output.dense", "down", li)
    m.set_weight("final.layernorm.gamma", fp["final_norm"], ft)
    m.set_weight("lm_head.weight", fp["lm_head"], ft)


def to_transformer_proto(graph, gen_ops=("GenerateOp", "UpdateId")):
    """The list as a SERIALIZED allspark TransformerProto (graph_proto.py: csrc/proto/allspark.proto), graphs "decoder" + "gen_graph"
    like the converter's export (qwen_v15.py:408-452) -- the bytes hostapi.Model.graph_add_serialized / the reference's AsModel read.
    Attributes: "k=i:v" -> int32, "f" -> float32, "b" -> one byte (the raw bytes InitV2 reads through a pointer cast)."""
    import struct
    from . import graph_proto as gp
    m = gp.TransformerProto()
    m.model_type = "Qwen_v15"
    m.graph_names.extend(["decoder", "gen_graph"])
    for t, name, inputs, outputs, weights, attrs in graph:
        op = m.graphs["gen_graph" if t in gen_ops else "decoder"].ops.add()
        op.op_type, op.op_name = t, name
        for lst, names in ((op.inputs, inputs), (op.outputs, outputs), (op.weights, weights)):
            for n in names:
                lst.add().name = n
        for kv in filter(None, attrs.split(";")):
            k, _, tv = kv.partition("=")
            ty, _, v = tv.partition(":")
            op.attr[k] = struct.pack("<i", int(v)) if ty == "i" else struct.pack("<f", float(v)) if ty == "f" else bytes([int(v) != 0])
    return m.SerializeToString()


def add_graph(m, graph):
    for t, name, inputs, outputs, weights, attrs in graph:
        m.graph_add_op(t, name, inputs, outputs, weights, attrs)
