// fusion_pass.cpp -- see fusion_pass.h.
#include "fusion_pass.h"

#include <cstring>

namespace allspark {

namespace {

const char* attr_ptr(const OperatorProto& p, const char* k) {
  auto it = p.attr.find(k);
  return it == p.attr.end() ? nullptr : it->second.c_str();
}
int attr_int(const OperatorProto& p, const char* k, int dflt) {
  const char* a = attr_ptr(p, k);
  return a ? *(const int*)a : dflt;
}
float attr_float(const OperatorProto& p, const char* k, float dflt) {
  const char* a = attr_ptr(p, k);
  return a ? *(const float*)a : dflt;
}
bool attr_bool(const OperatorProto& p, const char* k) {
  const char* a = attr_ptr(p, k);
  return a && *(const bool*)a;
}
std::string bytes_of(int v) { return std::string(reinterpret_cast<const char*>(&v), sizeof(v)); }

int lowp_bits(const OperatorProto& p) { return p.op_type == "GemmA16W8" ? 8 : p.op_type == "GemmA16W4" ? 4 : 0; }

struct Matcher {
  const std::vector<OperatorProto>& g;
  size_t i = 0;
  std::string why;
  bool fail(const std::string& w) {
    if (why.empty()) why = w + (i < g.size() ? " at operator #" + std::to_string(i) + " (" + g[i].op_type + " " + g[i].op_name + ")" : " at the end of the list");
    return false;
  }
  const OperatorProto* peek(size_t k = 0) const { return i + k < g.size() ? &g[i + k] : nullptr; }
  // a weight-only Gemm that reads `in`, with the expected activation, no residual fused in, plain alpha / layout
  bool lowp(const OperatorProto*& out, const std::string& in, int act, bool bias_ok) {
    const OperatorProto* p = peek();
    if (!p || !lowp_bits(*p)) return fail("expected GemmA16W8 / GemmA16W4");
    if (p->inputs.size() != 1 || p->inputs[0] != in || p->outputs.size() != 1) return fail("weight-only Gemm with unexpected inputs");
    if (p->weights.size() != 3 && !(bias_ok && p->weights.size() == 4)) return fail("weight-only Gemm with unexpected weights (bias)");
    if (attr_int(*p, "activation", 0) != act) return fail("weight-only Gemm with unexpected activation");
    if (attr_float(*p, "alpha", 1.0f) != 1.0f || attr_bool(*p, "transB") || attr_bool(*p, "is_pooler")) return fail("weight-only Gemm attributes");
    out = p;
    ++i;
    return true;
  }
  // an unquantised Gemm on `in` with one weight (no bias), the given activation, plain alpha / layout
  bool dense(const OperatorProto*& out, const std::string& in, int act) {
    const OperatorProto* p = peek();
    if (!p || p->op_type != "Gemm") return fail("expected Gemm");
    if (p->inputs.size() != 1 || p->inputs[0] != in || p->outputs.size() != 1 || p->weights.size() != 1) return fail("Gemm with unexpected inputs / weights");
    if (attr_int(*p, "activation", 0) != act || attr_bool(*p, "with_bias") || attr_bool(*p, "splitk") || attr_float(*p, "alpha", 1.0f) != 1.0f ||
        attr_bool(*p, "transB") || attr_int(*p, "binary_type", 0) != 0)
      return fail("Gemm attributes");
    out = p;
    ++i;
    return true;
  }
  // (further inputs are allowed: the exported graphs hand Rotary the position mask and GenerateOp the original ids;
  // several_outputs: GenerateOp also declares next_beam_idx / hyps outputs, model_base.py GenerateOp)
  bool typed(const OperatorProto*& out, const char* type, const std::string& in, bool several_outputs = false) {
    const OperatorProto* p = peek();
    if (!p || p->op_type != type) return fail(std::string("expected ") + type);
    if (p->inputs.empty() || p->inputs[0] != in || (several_outputs ? p->outputs.empty() : p->outputs.size() != 1))
      return fail(std::string(type) + " with unexpected inputs");
    out = p;
    ++i;
    return true;
  }
  // optional AllReduce(in) -> returns the name the sum ends up under
  std::string maybe_allreduce(const OperatorProto*& ar, const std::string& in) {
    ar = nullptr;
    const OperatorProto* p = peek();
    if (p && p->op_type == "AllReduce" && p->inputs.size() == 1 && p->inputs[0] == in && p->outputs.size() == 1) {
      ar = p;
      ++i;
      return p->outputs[0];
    }
    return in;
  }
  bool binary(const OperatorProto*& out, int type, const std::string& a, const std::string& b) {
    const OperatorProto* p = peek();
    if (!p || p->op_type != "Binary" || attr_int(*p, "binary_type", 0) != type) return fail("expected Binary " + std::string(type == 1 ? "ADD" : "MUL"));
    if (p->inputs.size() != 2 || p->outputs.size() != 1) return fail("Binary with unexpected inputs");
    const bool fwd = p->inputs[0] == a && p->inputs[1] == b, rev = p->inputs[0] == b && p->inputs[1] == a;
    if (!fwd && !rev) return fail("Binary does not combine the expected tensors");
    out = p;
    ++i;
    return true;
  }
};

OperatorProto make(const char* type, const std::string& name, std::vector<std::string> in, std::vector<std::string> out,
                   std::vector<std::string> w) {
  OperatorProto p;
  p.op_type = type;
  p.op_name = name;
  p.inputs = std::move(in);
  p.outputs = std::move(out);
  p.weights = std::move(w);
  return p;
}
void copy_attr(OperatorProto& dst, const OperatorProto& src, const char* k) {
  auto it = src.attr.find(k);
  if (it != src.attr.end()) dst.attr[k] = it->second;
}
void lowp_attrs(OperatorProto& dst, const OperatorProto& gemm) {
  dst.attr["wbits"] = bytes_of(lowp_bits(gemm));
  copy_attr(dst, gemm, "GroupSize");
}

}  // namespace

std::vector<OperatorProto> FuseDecoderGraph(const std::vector<OperatorProto>& graph, const DeviceContext& ctx, FusionReport* report) {
  FusionReport local;
  FusionReport& rep = report ? *report : local;
  rep = FusionReport();
  rep.ops_before = (int)graph.size();
  rep.ops_after = (int)graph.size();
  std::vector<OperatorProto> out;
  Matcher m{graph};
  auto refuse = [&](const std::string& w) {
    rep.why = w.empty() ? m.why : w;
    return graph;
  };
  if (ctx.GetDeviceType() != DeviceType::HIP) return refuse("not a HIP context");
  const OperatorProto* emb = m.peek();
  if (!emb || emb->op_type != "EmbeddingT5" || emb->inputs.size() != 1 || emb->outputs.size() != 1 || emb->weights.size() != 1)
    return refuse("the list does not start with EmbeddingT5(ids; word_embeddings)");
  ++m.i;
  out.push_back(make("DihipEmbedding", emb->op_name, emb->inputs, emb->outputs, emb->weights));
  std::string h = emb->outputs[0], xnorm_in;
  // the converter puts RichEmbedding(input_ids, embedding.out) -> embedding.out behind the embedding (qwen_v15.py:197-206: rows of
  // multimedia placeholders are overwritten IN PLACE with the request's own embeddings).  Without such inputs -- the model runner
  // has no path that carries them -- it is the identity: the fused list, whose embedding rows are the f32 hidden stream, drops it.
  if (const OperatorProto* re = m.peek(); re && re->op_type == "RichEmbedding") {
    if (re->inputs.size() != 2 || re->inputs[1] != h || re->outputs.size() != 1 || re->outputs[0] != h)
      return refuse(m.fail("RichEmbedding that is not the in-place form behind the embedding") ? "" : "");
    ++m.i;
    rep.why = "RichEmbedding dropped (identity without multimedia inputs; the fused list does not serve them)";
  }
  for (;;) {
    // a decoder layer starts with LayerNormNoBeta(h) followed by a weight-only Gemm; the tail with LayerNormNoBeta , GetLastLine
    const OperatorProto* ln1 = m.peek();
    const OperatorProto* nxt = m.peek(1);
    if (!ln1 || ln1->op_type != "LayerNormNoBeta" || ln1->inputs.size() != 1 || ln1->inputs[0] != h || ln1->weights.size() != 1 ||
        !attr_ptr(*ln1, "eps"))
      return refuse(m.fail("expected LayerNormNoBeta on the hidden rows") ? "" : "");
    if (!nxt || !lowp_bits(*nxt)) break;  // the tail
    ++m.i;
    const OperatorProto *qkv, *rot, *att, *o, *ar1, *add1, *ln2, *gate, *up, *mul, *down, *ar2, *add2;
    if (!m.lowp(qkv, ln1->outputs[0], 0, true)) return refuse("");
    if (!m.typed(rot, "Rotary", qkv->outputs[0])) return refuse("");
    if (attr_int(*rot, "rotary_type", 0) != 0 || attr_float(*rot, "rotary_pct", 1.0f) != 1.0f || attr_int(*rot, "invfreq_type", 0) != 0)
      return refuse("a Rotary variant the fused attention does not implement (" + rot->op_name + ")");
    for (const char* k : {"ntk_model_embed", "logn_model_embedding", "mrope_section_size", "seqlen_extrapolation", "rope_ratio",
                          "original_max_position_embeddings", "use_weight"}) {
      // (the converter always writes seqlen_extrapolation, 1.0 unless the user extrapolates: qwen_v15.py:228-230)
      if (std::string(k) == "seqlen_extrapolation" && attr_float(*rot, k, 1.0f) == 1.0f) continue;
      if (attr_ptr(*rot, k)) return refuse(std::string("Rotary attribute ") + k + " (" + rot->op_name + ")");
    }
    {
      const OperatorProto* p = m.peek();
      // (the converter appends GenerateOp's beam index to the attention's inputs, qwen_v15.py:445-447: unused by SpanAttnOp)
      if (!p || (p->op_type != "DecOptMQA" && p->op_type != "DecOptMHA") || p->inputs.empty() || p->inputs[0] != rot->outputs[0] ||
          p->outputs.size() != 1)
        return refuse(m.fail("expected DecOptMQA / DecOptMHA on the rotated rows") ? "" : "");
      att = p;
      ++m.i;
    }
    if (!m.lowp(o, att->outputs[0], 0, false)) return refuse("");
    const std::string o_sum = m.maybe_allreduce(ar1, o->outputs[0]);
    if (!m.binary(add1, 1, o_sum, h)) return refuse("");
    if (!m.typed(ln2, "LayerNormNoBeta", add1->outputs[0]) || ln2->weights.size() != 1 || !attr_ptr(*ln2, "eps")) return refuse(m.why.empty() ? "ffn LayerNormNoBeta" : "");
    const OperatorProto* after_ln2 = m.peek();
    const bool moe_layer = after_ln2 && after_ln2->op_type == "Gemm";

    OperatorProto f_qkv = make("DihipNormGemm", qkv->op_name, {h}, qkv->outputs, {ln1->weights[0]});
    if (!xnorm_in.empty()) f_qkv.inputs.push_back(xnorm_in);
    f_qkv.weights.insert(f_qkv.weights.end(), qkv->weights.begin(), qkv->weights.end());
    copy_attr(f_qkv, *ln1, "eps");
    lowp_attrs(f_qkv, *qkv);
    out.push_back(std::move(f_qkv));

    OperatorProto f_att = make("DihipRopeSpanAttn", att->op_name, {qkv->outputs[0]}, att->outputs, {});
    f_att.attr = rot->attr;
    for (const auto& kv : att->attr) f_att.attr[kv.first] = kv.second;
    out.push_back(std::move(f_att));
    xnorm_in.clear();

    if (moe_layer) {
      // the mixture-of-experts feed-forward block (qwen_v20_moe.py:318-382) -> one DihipMoeBlock
      const OperatorProto *router, *moe, *ar_moe, *sgu, *glu, *sdown, *sgate, *calc, *ar_sh, *eadd;
      if (!m.dense(router, ln2->outputs[0], 0)) return refuse("");
      {
        const OperatorProto* p = m.peek();
        if (!p || p->op_type != "MOEA16W8" || p->inputs.size() != 2 || p->inputs[0] != ln2->outputs[0] || p->inputs[1] != router->outputs[0] ||
            p->outputs.size() != 1 || p->weights.size() != 6)
          return refuse(m.fail("expected MOEA16W8(ffn rows, router logits)") ? "" : "");
        moe = p;
        ++m.i;
      }
      (void)m.maybe_allreduce(ar_moe, moe->outputs[0]);
      if (!m.lowp(sgu, ln2->outputs[0], 0, false)) return refuse("");
      if (!m.typed(glu, "UnaryGLU", sgu->outputs[0])) return refuse("");
      if (attr_int(*glu, "unary_type", 0) != (int)SILU) return refuse("a UnaryGLU activation the fused shared expert does not implement (" + glu->op_name + ")");
      if (!m.lowp(sdown, glu->outputs[0], 0, false)) return refuse("");
      if (lowp_bits(*sgu) != lowp_bits(*sdown) || attr_int(*sgu, "GroupSize", -1) != attr_int(*sdown, "GroupSize", -1))
        return refuse("shared expert projections quantised differently (" + sgu->op_name + ")");
      if (!m.dense(sgate, ln2->outputs[0], (int)SIGMOID)) return refuse("");
      {
        const OperatorProto* p = m.peek();
        if (!p || p->op_type != "CalcExpert" || p->inputs.size() != 2 || p->inputs[0] != sdown->outputs[0] || p->inputs[1] != sgate->outputs[0] ||
            p->outputs.size() != 1)
          return refuse(m.fail("expected CalcExpert(shared expert rows, shared expert gate)") ? "" : "");
        calc = p;
        ++m.i;
      }
      (void)m.maybe_allreduce(ar_sh, calc->outputs[0]);
      if (!m.binary(eadd, 1, moe->outputs[0], calc->outputs[0])) return refuse("");
      if (!m.binary(add2, 1, eadd->outputs[0], add1->outputs[0])) return refuse("");

      OperatorProto f_o = make("DihipGemmAddTo", o->op_name, {att->outputs[0], h}, {add1->outputs[0]}, o->weights);
      lowp_attrs(f_o, *o);
      out.push_back(std::move(f_o));
      if (ar1) out.push_back(make("AllReduce", ar1->op_name, {add1->outputs[0]}, {add1->outputs[0]}, {}));

      OperatorProto f_moe = make("DihipMoeBlock", moe->op_name, {add1->outputs[0]}, {add2->outputs[0]}, {ln2->weights[0], router->weights[0]});
      for (const OperatorProto* src : {moe, sgu, sdown}) f_moe.weights.insert(f_moe.weights.end(), src->weights.begin(), src->weights.end());
      f_moe.weights.push_back(sgate->weights[0]);
      copy_attr(f_moe, *ln2, "eps");
      for (const char* k : {"num_experts", "num_experts_per_tok", "use_ep", "GroupSize"}) copy_attr(f_moe, *moe, k);
      f_moe.attr["wbits"] = bytes_of(lowp_bits(*sgu));
      if (attr_ptr(*sgu, "GroupSize")) f_moe.attr["shared.GroupSize"] = sgu->attr.at("GroupSize");
      out.push_back(std::move(f_moe));
      // the reference all-reduces the MOE rows and the CalcExpert rows, the block adds them (and the residual on rank 0) first:
      // one all-reduce of the f32 hidden rows, the same sum
      // -- which is only the same sum when BOTH partial results are rank-partial (or neither): one all-reduce present without
      // the other means one addend is replicated and would be counted nranks times
      if ((ar_moe != nullptr) != (ar_sh != nullptr))
        return refuse("mixture-of-experts block: one of the two all-reduces (experts / shared expert) is missing (" + moe->op_name + ")");
      if (ar_moe || ar_sh) out.push_back(make("AllReduce", (ar_moe ? ar_moe : ar_sh)->op_name, {add2->outputs[0]}, {add2->outputs[0]}, {}));
      h = add2->outputs[0];
      ++rep.layers;
      continue;
    }

    if (!m.lowp(gate, ln2->outputs[0], (int)SILU, false)) return refuse("");
    if (!m.lowp(up, ln2->outputs[0], 0, false)) return refuse("");
    if (lowp_bits(*gate) != lowp_bits(*up) || attr_int(*gate, "GroupSize", -1) != attr_int(*up, "GroupSize", -1))
      return refuse("gate / up projections quantised differently (" + gate->op_name + ")");
    if (!m.binary(mul, 2, gate->outputs[0], up->outputs[0])) return refuse("");
    if (!m.lowp(down, mul->outputs[0], 0, false)) return refuse("");
    const std::string d_sum = m.maybe_allreduce(ar2, down->outputs[0]);
    if (!m.binary(add2, 1, d_sum, add1->outputs[0])) return refuse("");
    // the norm that follows the layer (the next layer's first, never the final one: the lm_head fuses that itself)
    const OperatorProto* next_ln = m.peek();
    const OperatorProto* next_gemm = m.peek(1);
    const bool hand_on = next_ln && next_ln->op_type == "LayerNormNoBeta" && next_ln->inputs.size() == 1 && next_ln->inputs[0] == add2->outputs[0] &&
                         next_ln->weights.size() == 1 && attr_ptr(*next_ln, "eps") && next_gemm && lowp_bits(*next_gemm);

    const std::string xn2 = o->op_name + ".dihip_xnorm";
    OperatorProto f_o = make("DihipGemmAddTo", o->op_name, {att->outputs[0], h}, {add1->outputs[0], xn2}, o->weights);
    f_o.weights.push_back(ln2->weights[0]);
    copy_attr(f_o, *ln2, "eps");
    lowp_attrs(f_o, *o);
    out.push_back(std::move(f_o));
    if (ar1) out.push_back(make("AllReduce", ar1->op_name, {add1->outputs[0]}, {add1->outputs[0]}, {}));

    OperatorProto f_mlp = make("DihipNormSwiGLU", gate->op_name, {add1->outputs[0], xn2}, mul->outputs, {ln2->weights[0]});
    f_mlp.weights.insert(f_mlp.weights.end(), gate->weights.begin(), gate->weights.end());
    f_mlp.weights.insert(f_mlp.weights.end(), up->weights.begin(), up->weights.end());
    copy_attr(f_mlp, *ln2, "eps");
    lowp_attrs(f_mlp, *gate);
    out.push_back(std::move(f_mlp));

    OperatorProto f_down = make("DihipGemmAddTo", down->op_name, {mul->outputs[0], add1->outputs[0]}, {add2->outputs[0]}, down->weights);
    lowp_attrs(f_down, *down);
    if (hand_on) {
      xnorm_in = down->op_name + ".dihip_xnorm";
      f_down.outputs.push_back(xnorm_in);
      f_down.weights.push_back(next_ln->weights[0]);
      copy_attr(f_down, *next_ln, "eps");
    }
    out.push_back(std::move(f_down));
    if (ar2) out.push_back(make("AllReduce", ar2->op_name, {add2->outputs[0]}, {add2->outputs[0]}, {}));
    h = add2->outputs[0];
    ++rep.layers;
  }
  if (rep.layers == 0) return refuse("no decoder layer matched");
  // tail: LayerNormNoBeta(h) , GetLastLine , Gemm(lm_head.weight) , GenerateOp
  const OperatorProto *lnf = m.peek(), *gll, *lm, *gen;
  ++m.i;
  const size_t tail_at = m.i;
  bool dropped_update_id = false;
  auto simple_tail = [&]() -> bool {
    if (!m.typed(gll, "GetLastLine", lnf->outputs[0])) return false;
    if (!m.typed(lm, "Gemm", gll->outputs[0])) return false;
    if (lm->weights.size() != 1 || attr_bool(*lm, "splitk") || attr_bool(*lm, "with_bias") || attr_int(*lm, "activation", 0) != 0 ||
        attr_float(*lm, "alpha", 1.0f) != 1.0f || attr_int(*lm, "binary_type", 0) != 0 || lm->inputs.size() != 1)
      return m.fail("an lm_head Gemm the fused head does not implement");
    if (ctx.GetNranks() > 1) return m.fail("tensor-parallel lm_head: the fused head is single-rank");
    if (m.i == graph.size()) {  // the decoder graph alone (AsModel keeps GenerateOp in gen_graph): fused up to the f32 logits
      gen = nullptr;
      return true;
    }
    if (!m.typed(gen, "GenerateOp", lm->outputs[0], true)) return false;
    // gen_graph = [GenerateOp, UpdateId] (qwen_v15.py:436-448): the stop checks of UpdateId need the token on the host; in the
    // fused list they are the model runner's, at its sync points (HipModelRunner::Sync / dihost_request_poll)
    if (const OperatorProto* u = m.peek(); u && u->op_type == "UpdateId" && m.i + 1 == graph.size()) {
      dropped_update_id = true;
      ++m.i;
    }
    if (m.i != graph.size()) return m.fail("operators after GenerateOp");
    return true;
  };
  if (!simple_tail()) {
    // the tail as it is, behind a DihipFinalNorm: every remaining operator must be one the HIP backend registers and that reads FT rows
    const std::string why_simple = m.why;
    size_t tail_end = graph.size();
    if (tail_end > tail_at + 1 && graph[tail_end - 1].op_type == "UpdateId" && graph[tail_end - 2].op_type == "GenerateOp") --tail_end;  // as above
    for (size_t j = tail_at; j < tail_end; ++j) {
      const std::string& t = graph[j].op_type;
      if (t != "GetLastLine" && t != "Gemm" && t != "AllReduce" && t != "AllGather" && t != "GenerateOp")
        return refuse(why_simple + ", and the tail holds " + t + " (" + graph[j].op_name + "), which cannot stay behind DihipFinalNorm");
    }
    if (tail_at >= tail_end || graph[tail_end - 1].op_type != "GenerateOp") return refuse(why_simple + ", and the list does not end in GenerateOp");
    OperatorProto f_norm = make("DihipFinalNorm", lnf->op_name, {h}, lnf->outputs, lnf->weights);
    copy_attr(f_norm, *lnf, "eps");
    out.push_back(std::move(f_norm));
    for (size_t j = tail_at; j < tail_end; ++j) out.push_back(graph[j]);
    // the sampling operator of such a tail: DihipGreedy reads FT logits too (cast to f32 first) and advances the device-resident
    // length counters in its launch, so the step state stays on the device under tensor parallelism as well -- GetLastLine, Gemm(splitk)
    // and AllReduce only enqueue -- and the step replays as a hipGraph.  Sampling parameters as in the one-rank list.
    OperatorProto& gen_tp = out.back();
    gen_tp.op_type = "DihipGreedy";
    gen_tp.inputs.resize(1);
    rep.fused = true;
    rep.device_resident = true;
    rep.why = "layers fused, the tail keeps the reference's operators (" + why_simple + ") in front of DihipGreedy";
    rep.ops_after = (int)out.size();
    return out;
  }
  OperatorProto f_lm = make("DihipLMHead", lm->op_name, {h}, lm->outputs, {lnf->weights[0], lm->weights[0]});
  copy_attr(f_lm, *lnf, "eps");
  out.push_back(std::move(f_lm));
  if (!gen) {
    // GenerateOp (gen_graph) takes the f32 logits as they are (host/glue_ops_hip.cpp); without DihipGreedy nothing advances the
    // device-resident length counters: the step state is staged per step, as with an unfused tail
    rep.fused = true;
    rep.device_resident = false;
    rep.why = "no GenerateOp in this list (the decoder graph alone): fused up to the f32 logits, step state staged from the host";
    rep.ops_after = (int)out.size();
    return out;
  }
  OperatorProto f_gen = make("DihipGreedy", gen->op_name, {lm->outputs[0]}, gen->outputs, {});
  f_gen.attr = gen->attr;
  out.push_back(std::move(f_gen));
  if (dropped_update_id) rep.why = "UpdateId behind GenerateOp is left to the model runner's sync points";
  rep.fused = true;
  rep.device_resident = true;
  rep.ops_after = (int)out.size();
  return out;
}

}  // namespace allspark
