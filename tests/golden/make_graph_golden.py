#!/usr/bin/env python3
"""tests/golden/qwen2_a16w4_g128.asgraph.pb: a SERIALIZED `TransformerProto` (csrc/proto/allspark.proto) whose graphs were built
by the REFERENCE'S OWN converter -- `Qwen_v15._build_graph` of python/pyhie/allspark/model/qwen_v15.py:33-452, with the reference's
`Operator` classes (model_base.py:52-400) and its weight-only rewrite `quantize_op` (quantization_utils.py:56-110) -- run in this
container over message classes that dash-infer_amd/graph_proto.py builds from the .proto's field table (the generated
`_allspark` extension the package imports is a compiled module that does not exist here: a stand-in module exports those message
classes, the enum constants and nothing else).

Only the GRAPHS are produced (weights: none -- `_trans_weight` is not run; the tests bind random weights by the names the graph
declares).  Runs only where /root/reference exists; the committed bytes travel.  No reference source is copied.

    python tests/golden/make_graph_golden.py        # rewrites tests/golden/qwen2_a16w4_g128.asgraph.pb (+ the int8 per-channel one)
"""
import importlib
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.environ.get("DASHINFER_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def install_reference_package():
    """pyhie.allspark(.model) as bare namespace modules over the reference's directories (their __init__.py would import the
    engine), and pyhie.allspark._allspark as the stand-in described above."""
    from tests.conftest import load_pkg
    load_pkg()
    from dash_infer_amd import graph_proto as gp
    base = os.path.join(REF, "python", "pyhie")
    for name, path in (("pyhie", base), ("pyhie.allspark", os.path.join(base, "allspark")),
                       ("pyhie.allspark.model", os.path.join(base, "allspark", "model")),
                       ("pyhie.allspark.quant", os.path.join(base, "allspark", "quant"))):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    stub = types.ModuleType("pyhie.allspark._allspark")
    for cls in ("TensorProto", "OperatorProto", "GraphProto", "TransformerProto", "ConfigProto", "BuildVersion", "WeightHash", "BuildMetaProto"):
        setattr(stub, cls, getattr(gp, cls))
    for vals in gp.ENUMS.values():
        for k, v in vals.items():
            setattr(stub, k, v)
    stub.__all__ = [k for k in vars(stub) if not k.startswith("__")]
    sys.modules["pyhie.allspark._allspark"] = stub
    return gp


def build(gp, hf_cfg, weight_type, subchannel, group):
    mb = importlib.import_module("pyhie.allspark.model.model_base")
    # (model_base.make_tensor serialises array data with trans_to_allsparkz -> the compiled save_allsparky: the graph's two input
    # declarations carry an empty int64 array there; the tensor NAME is what the graph needs)
    mb.trans_to_allsparkz = lambda data, *a, **k: b""
    qmod = importlib.import_module("pyhie.allspark.model.qwen_v15")
    quantization = importlib.import_module("pyhie.allspark.quantization")
    QC = quantization.QuantizeConfig
    qc = QC.__new__(QC)                       # (its __init__ wants the user-facing QuantizationSettings: the fields the graph
    qc.weight_type = weight_type              # builder reads are set directly, as load_GPTQ_config does)
    qc.extra_option = {"SubChannel": subchannel, **({"GroupSize": group} if subchannel else {})}
    qc.quantize_mode = QC.QuantMode.A16W4 if weight_type == "UINT4" else QC.QuantMode.A16W8
    names = ["model.embed_tokens.weight", "model.norm.weight", "lm_head.weight"]
    for i in range(hf_cfg["num_hidden_layers"]):
        p = f"model.layers.{i}."
        names += [p + s for s in ("self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight", "self_attn.o_proj.weight",
                                  "self_attn.q_proj.bias", "self_attn.k_proj.bias", "self_attn.v_proj.bias", "input_layernorm.weight",
                                  "post_attention_layernorm.weight", "mlp.gate_proj.weight", "mlp.up_proj.weight", "mlp.down_proj.weight")]
    m = qmod.Qwen_v15.__new__(qmod.Qwen_v15)
    mb.Model.__init__(m, "Qwen_v15", "bfloat16", hf_cfg, multigpu_mode=0, do_binary_add_fused=True, do_dynamic_quantize_convert=True,
                      quant_config=qc, rotary_base=hf_cfg["rope_theta"])
    # what Qwen_v15.__init__ does before _build_graph (qwen_v15.py:13-27), minus the weight conversion
    m.model.inputs.append(mb.make_tensor("input_ids", np.empty((0, 0), np.int64)))
    m.model.inputs.append(mb.make_tensor("attention_mask", np.empty((0, 0), np.int64)))
    m.model.outputs.append(mb.make_tensor("last_hidden_state"))
    m.is_generate = True
    m.weight_real_names = set(names)
    m._build_graph(hf_cfg, "lmhead")
    return m.model


def main():
    gp = install_reference_package()
    hf = {"rms_norm_eps": 1e-6, "num_attention_heads": 28, "num_key_value_heads": 4, "num_hidden_layers": 2, "hidden_size": 3584,
          "hidden_act": "silu", "size_per_head": 128, "intermediate_size": 1024, "rope_theta": 1000000.0}
    for fname, wt, sub, grp in (("qwen2_a16w4_g128.asgraph.pb", "UINT4", True, 128), ("qwen2_a16w8_perc.asgraph.pb", "INT8", False, -1)):
        model = build(gp, hf, wt, sub, grp)
        data = model.SerializeToString()
        open(os.path.join(OUT, fname), "wb").write(data)
        print(fname, len(data), "bytes; graphs:", list(model.graph_names))
        for gname in model.graph_names:
            ops = model.graphs[gname].ops
            print(" ", gname, len(ops), "ops:", " ".join(o.op_type for o in ops[:12]), "..." if len(ops) > 12 else "")
        dec = model.graphs["decoder"].ops
        for o in dec[:14]:
            print("    ", o.op_type, o.op_name, [t.name for t in o.inputs], "->", [t.name for t in o.outputs], [t.name for t in o.weights],
                  {k: (v.hex() if len(v) <= 8 else v) for k, v in o.attr.items()})


if __name__ == "__main__":
    main()
