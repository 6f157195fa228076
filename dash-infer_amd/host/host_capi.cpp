// host_capi.cpp -- C test harness of the operator layer (include/dashinfer_hip_host.h).
#include "dashinfer_hip_host.h"

#include <cstring>
#include <mutex>
#include <sstream>
#include <thread>

#include <algorithm>

#include "graph_wire.h"
#include "weight_file.h"
#include "model_runner.h"
#include "operator.h"

using namespace allspark;

// Test stand-in for SpannedVirtualCache (csrc/runtime/cache/virtual_cache.{h,cpp}): every layer owns a pre-assigned list of
// span pointers (the harness hands them in, so that tests know which span holds what) and claims them in order as the
// sequence grows.  Thread-safe per object, like the reference's (the model allocates the layers concurrently).
class ListVirtualCache : public VirtualCache {
 public:
  ListVirtualCache(std::vector<std::vector<void*>> spans, int span_len, size_t initial_len)
      : spans_(std::move(spans)), span_len_(span_len), len_(spans_.size(), initial_len) {
    for (size_t l = 0; l < spans_.size(); ++l) {
      views_.push_back(nullptr);
      refresh(l);
    }
  }
  const AsTensor& GetCache(int layer, int increment) override {
    std::lock_guard<std::mutex> g(mu_);
    if (layer < 0 || layer >= (int)spans_.size() || increment < 0) throw AsException("GetCache: bad argument", AsStatus::ALLSPARK_PARAM_ERROR);
    const size_t new_len = len_[layer] + (size_t)increment;
    if ((new_len + span_len_ - 1) / span_len_ > spans_[layer].size())
      throw AsException("GetCache: no span left", AsStatus::ALLSPARK_CACHE_MEMORY_OUT);
    len_[layer] = new_len;
    refresh(layer);
    ++calls_;
    return *views_[layer];
  }
  void Rewind(size_t len) {  // benchmarks: the cache is taken back to `len` tokens (its spans stay assigned)
    std::lock_guard<std::mutex> g(mu_);
    for (size_t l = 0; l < len_.size(); ++l) {
      len_[l] = len;
      refresh(l);
    }
  }
  size_t GetSeqLength(int layer) const override {
    std::lock_guard<std::mutex> g(mu_);
    return layer >= 0 && layer < (int)len_.size() ? len_[layer] : 0;
  }
  int GetLayerNum() const override { return (int)spans_.size(); }
  long calls() const { return calls_; }

 private:
  void refresh(size_t l) {  // the POINTER tensor covers the spans claimed so far
    const int64_t n = (int64_t)((len_[l] + span_len_ - 1) / span_len_);
    views_[l] = std::make_unique<AsTensor>("span_ptrs", DeviceType::CPU, POINTER, Shape{n}, spans_[l].data());
  }
  std::vector<std::vector<void*>> spans_;
  int span_len_;
  std::vector<size_t> len_;
  std::vector<std::unique_ptr<AsTensor>> views_;
  mutable std::mutex mu_;
  long calls_ = 0;
};

struct dihost_model {
  HIPContext ctx;
  TensorMap tensors, weights, weights_buffer;
  RuntimeContext rt;
  std::vector<std::unique_ptr<AsOperator>> ops;
  std::vector<OperatorProto> graph;          // dihost_graph_add_op
  std::unique_ptr<HipModelRunner> runner;    // dihost_graph_build
  std::string text;
  GenerateConfig next_gen;                   // dihost_next_request_generation
  bool has_next_gen = false;
};
static thread_local std::string g_err;

static std::vector<std::string> split(const char* s, char sep) {
  std::vector<std::string> out;
  if (!s) return out;
  std::stringstream ss(s);
  std::string item;
  while (std::getline(ss, item, sep))
    if (!item.empty()) out.push_back(item);
  return out;
}

extern "C" {

const char* dihost_last_error(void) { return g_err.c_str(); }

const char* dihost_registered_ops(void) {
  static std::string s;
  s.clear();
  for (const char* t : {"GemmA16W8", "GemmA16W4", "DecOptMHA", "DecOptMQA", "AllReduce", "AllGather", "MOEA16W8", "CalcExpert", "Gemm", "Rotary",
                        "LayerNormNoBeta", "Binary", "Unary", "UnaryGLU", "EmbeddingT5", "GetLastLine", "GenerateOp", "TransMask", "RichEmbedding",
                        "PreProcessId", "UpdateId", "PostProcessId", "DihipEmbedding", "DihipNormGemm", "DihipRopeSpanAttn", "DihipGemmAddTo",
                        "DihipNormSwiGLU", "DihipLMHead", "DihipGreedy", "DihipFinalNorm", "DihipMoeBlock"}) {
    try {
      (void)OpFactory::getInstance().GetOperator({t, DeviceType::HIP});
      s += (s.empty() ? "" : ",");
      s += t;
    } catch (const AsException&) {
    }
  }
  return s.c_str();
}

int dihost_model_create(dihost_model_t* m, void* stream, int num_heads, int num_groups, int size_per_head, int span_size,
                        int cache_mode, int max_batch, int max_length, int rank, int nranks, void* rccl_comm) {
  if (!m) return (int)AsStatus::ALLSPARK_PARAM_ERROR;
  auto* p = new dihost_model();
  p->ctx.SetStream(reinterpret_cast<hipStream_t>(stream));
  p->ctx.SetNumberHeads(num_heads);
  p->ctx.SetNumberGroups(num_groups);
  p->ctx.SetSizePerHead(size_per_head);
  p->ctx.SetCacheSpanSize(span_size);
  p->ctx.SetCacheMode(static_cast<AsCacheMode>(cache_mode));
  p->ctx.SetModelMaxBatch(max_batch);
  p->ctx.SetModelMaxLength(max_length);
  p->ctx.SetRankInfo(rank, nranks);
  p->ctx.SetRCCLComm(rccl_comm);
  p->tensors.emplace("workspace", std::make_shared<AsTensor>("workspace", DeviceType::HIP, INT8, Shape{256}));
  *m = p;
  return 0;
}
int dihost_model_set_p2p_comm(dihost_model_t m, void* p2p_comm) {
  if (!m) return (int)AsStatus::ALLSPARK_PARAM_ERROR;
  m->ctx.SetP2PComm(p2p_comm);
  return 0;
}
int dihost_model_destroy(dihost_model_t m) {
  delete m;
  return 0;
}
static int put(TensorMap& map, const char* name, int dtype, int ndim, const int64_t* shape, void* data) {
  if (!name || ndim < 0 || ndim > 8) return (int)AsStatus::ALLSPARK_PARAM_ERROR;
  Shape s(shape, shape + ndim);
  map[name] = std::make_shared<AsTensor>(name, DeviceType::HIP, static_cast<DataType>(dtype), s, data);
  return 0;
}
int dihost_set_tensor(dihost_model_t m, const char* name, int dtype, int ndim, const int64_t* shape, void* data) {
  return put(m->tensors, name, dtype, ndim, shape, data);
}
int dihost_set_weight(dihost_model_t m, const char* name, int dtype, int ndim, const int64_t* shape, void* data) {
  return put(m->weights, name, dtype, ndim, shape, data);
}
int dihost_get_weight(dihost_model_t m, const char* name, int* dtype, int* ndim, int64_t* shape8, void** data) {
  auto it = m->weights.find(name);
  if (it == m->weights.end()) return (int)AsStatus::ALLSPARK_PARAM_ERROR;
  const AsTensor& t = *it->second;
  if (dtype) *dtype = t.GetDataType();
  if (ndim) *ndim = (int)t.GetShape().size();
  if (shape8)
    for (size_t i = 0; i < t.GetShape().size() && i < 8; ++i) shape8[i] = t.GetShape()[i];
  if (data) *data = t.GetDataPtr();
  if (!t.GetDataPtr() && t.Count() > 0) {  // (a released weight must not look like an empty one)
    g_err = std::string("weight ") + name + " was released after its operator re-laid it out (InitV2): type and shape are still valid, the data is gone";
    return (int)AsStatus::ALLSPARK_INVALID_CALL_ERROR;
  }
  return 0;
}
// the records of a serialized weight file (.asparam, host/weight_file.h) as text, one per line: name|dtype|d0,d1,...|split_mode|offset|nbytes
// (no device work: usable without a GPU).  *need = bytes of the whole text incl. the terminator; out may be null / short
int dihost_weight_file_index(const char* path, char* out, size_t cap, size_t* need) {
  std::vector<WeightRecord> recs;
  std::string err;
  if (!path || !IndexWeightFile(path, &recs, &err)) {
    g_err = err.empty() ? "weight file: null path" : err;
    return (int)AsStatus::ALLSPARK_PARAM_ERROR;
  }
  std::string text;
  for (const WeightRecord& r : recs) {
    text += r.name + "|" + std::to_string((int)r.dtype) + "|";
    for (size_t i = 0; i < r.shape.size(); ++i) text += (i ? "," : "") + std::to_string(r.shape[i]);
    text += "|" + std::to_string(r.split_mode) + "|" + std::to_string(r.offset) + "|" + std::to_string(r.nbytes) + "\n";
  }
  if (need) *need = text.size() + 1;
  if (out && cap > 0) {
    const size_t n = std::min(cap - 1, text.size());
    std::memcpy(out, text.data(), n);
    out[n] = 0;
  }
  return 0;
}
// every record of the file becomes a weight of the model under its own name, in device memory the model OWNS (the weight-only
// operators free it once they have re-laid it out, as the reference re-lays-out in place): what WeightManager::LoadWeightForModel
// hands the operators (csrc/runtime/weight/weight_manager.cpp), for one rank (the TP split is the converter's / the caller's)
int dihost_weights_load_file(dihost_model_t m, const char* path, int* count) {
  if (!m) return (int)AsStatus::ALLSPARK_PARAM_ERROR;
  std::vector<WeightRecord> recs;
  std::string err;
  if (!path || !IndexWeightFile(path, &recs, &err)) {
    g_err = err.empty() ? "weight file: null path" : err;
    return (int)AsStatus::ALLSPARK_PARAM_ERROR;
  }
  FILE* fp = std::fopen(path, "rb");
  if (!fp) return (int)AsStatus::ALLSPARK_IO_ERROR;
  // every record is split for THIS rank by its SplitMode on the way in (WeightManager::LoadWeightForModel -> WeightSplitter,
  // csrc/runtime/weight/weight_manager.cpp; weight_file.h SliceForRank): a converter export of the whole model feeds any TP degree
  const int rank = m->ctx.GetRank(), nranks = std::max(1, m->ctx.GetNranks());
  std::vector<char> stage, share;
  std::vector<int64_t> shape;
  int n = 0;
  for (const WeightRecord& r : recs) {
    if (!ReadRecordDense(fp, r, &stage, &err)) {  // (CSC / ELL records densified: weight_file.h)
      std::fclose(fp);
      g_err = "weight file: " + err;
      return (int)AsStatus::ALLSPARK_IO_ERROR;
    }
    if (!SliceForRank(r, stage.data(), rank, nranks, &share, &shape, &err)) {
      std::fclose(fp);
      g_err = "weight file: " + err;
      return (int)AsStatus::ALLSPARK_PARAM_ERROR;
    }
    auto t = std::make_shared<AsTensor>(r.name, DeviceType::HIP, r.dtype, Shape(shape.begin(), shape.end()));
    if (t->GetSizeInByte() < share.size() || (!share.empty() && !t->GetDataPtr())) {
      std::fclose(fp);
      g_err = "weight file: cannot allocate " + r.name;
      return (int)AsStatus::ALLSPARK_MEMORY_ERROR;
    }
    if (!share.empty() && hipMemcpy(t->GetDataPtr(), share.data(), share.size(), hipMemcpyHostToDevice) != hipSuccess) {
      std::fclose(fp);
      g_err = "weight file: cannot upload " + r.name;
      return (int)AsStatus::ALLSPARK_IO_ERROR;
    }
    m->weights[r.name] = t;
    ++n;
  }
  std::fclose(fp);
  if (count) *count = n;
  return 0;
}
// one record's share for (rank, nranks) on the HOST (no GPU, no model): what dihost_weights_load_file uploads for that rank.
// data == NULL: sizes only.  shape8 / ndim: the share's shape.
int dihost_weight_file_slice(const char* path, const char* name, int rank, int nranks, void* data, size_t capacity, size_t* nbytes, int64_t* shape8,
                             int* ndim) {
  std::vector<WeightRecord> recs;
  std::string err;
  if (!path || !name || !IndexWeightFile(path, &recs, &err)) {
    g_err = err.empty() ? "weight file: null argument" : err;
    return (int)AsStatus::ALLSPARK_PARAM_ERROR;
  }
  for (const WeightRecord& r : recs) {
    if (r.name != name) continue;
    FILE* fp = std::fopen(path, "rb");
    if (!fp) return (int)AsStatus::ALLSPARK_IO_ERROR;
    std::vector<char> stage, share;
    std::vector<int64_t> shape;
    const bool ok = ReadRecordDense(fp, r, &stage, &err);
    std::fclose(fp);
    if (!ok) {
      g_err = "weight file: " + err;
      return (int)AsStatus::ALLSPARK_IO_ERROR;
    }
    if (!SliceForRank(r, stage.data(), rank, std::max(1, nranks), &share, &shape, &err)) {
      g_err = "weight file: " + err;
      return (int)AsStatus::ALLSPARK_PARAM_ERROR;
    }
    if (nbytes) *nbytes = share.size();
    if (ndim) *ndim = (int)shape.size();
    if (shape8)
      for (size_t i = 0; i < shape.size() && i < 8; ++i) shape8[i] = shape[i];
    if (data) {
      if (capacity < share.size()) return (int)AsStatus::ALLSPARK_MEMORY_ERROR;
      std::memcpy(data, share.data(), share.size());
    }
    return 0;
  }
  g_err = std::string("weight file: no record named ") + name;
  return (int)AsStatus::ALLSPARK_PARAM_ERROR;
}
int dihost_get_tensor(dihost_model_t m, const char* name, int* dtype, int* ndim, int64_t* shape8, void** data) {
  auto it = m->tensors.find(name);
  if (it == m->tensors.end()) return (int)AsStatus::ALLSPARK_PARAM_ERROR;
  const AsTensor& t = *it->second;
  if (dtype) *dtype = t.GetDataType();
  if (ndim) *ndim = (int)t.GetShape().size();
  if (shape8)
    for (size_t i = 0; i < t.GetShape().size() && i < 8; ++i) shape8[i] = t.GetShape()[i];
  if (data) *data = t.GetDataPtr();
  return 0;
}

static int parse_proto(OperatorProto& proto, const char* op_type, const char* op_name, const char* inputs, const char* outputs,
                       const char* weights, const char* attrs);

int dihost_op_create(dihost_model_t m, int* op_id, const char* op_type, const char* op_name, const char* inputs,
                     const char* outputs, const char* weights, const char* attrs) {
  OperatorProto proto;
  const int prc = parse_proto(proto, op_type, op_name, inputs, outputs, weights, attrs);
  if (prc) return prc;
  std::unique_ptr<AsOperator> op;
  try {
    op = OpFactory::getInstance().GetOperator({proto.op_type, DeviceType::HIP})();
  } catch (const AsException& e) {
    g_err = e.what();
    return (int)AsStatus::ALLSPARK_PARAM_ERROR;
  }
  const AsStatus st = op->CallInit(proto, m->ctx, m->weights, m->weights_buffer, &m->tensors, &m->rt);
  if (st != AsStatus::ALLSPARK_SUCCESS) {
    g_err = "CallInit failed";
    return (int)st;
  }
  m->ops.push_back(std::move(op));
  if (op_id) *op_id = (int)m->ops.size() - 1;
  return 0;
}

static int parse_proto(OperatorProto& proto, const char* op_type, const char* op_name, const char* inputs, const char* outputs,
                       const char* weights, const char* attrs) {
  proto.op_type = op_type ? op_type : "";
  proto.op_name = op_name ? op_name : "";
  proto.inputs = split(inputs, ',');
  proto.outputs = split(outputs, ',');
  proto.weights = split(weights, ',');
  for (const std::string& kv : split(attrs, ';')) {
    const size_t eq = kv.find('=');
    if (eq == std::string::npos || eq + 2 >= kv.size() || kv[eq + 2] != ':') {
      g_err = "bad attribute: " + kv;
      return (int)AsStatus::ALLSPARK_PARAM_ERROR;
    }
    const std::string key = kv.substr(0, eq), val = kv.substr(eq + 3);
    std::string bytes;
    if (kv[eq + 1] == 'i') {
      const int v = std::stoi(val);
      bytes.assign(reinterpret_cast<const char*>(&v), sizeof(v));
    } else if (kv[eq + 1] == 'f') {
      const float v = std::stof(val);
      bytes.assign(reinterpret_cast<const char*>(&v), sizeof(v));
    } else {
      const bool v = val != "0";
      bytes.assign(reinterpret_cast<const char*>(&v), sizeof(v));
    }
    proto.attr[key] = bytes;
  }
  return 0;
}

int dihost_set_runtime(dihost_model_t m, int is_context, int n_requests, const int* steps, int n_layers, int spans_per_req,
                       void* const* k_spans, void* const* v_spans) {
  m->rt.is_context = is_context != 0;
  m->rt.current_batch = 0;
  m->rt.gen_ctx_list.clear();
  for (int r = 0; r < n_requests; ++r) {
    auto gc = std::make_shared<GenerateContext>();
    gc->step = steps ? steps[r] : 0;
    std::vector<std::vector<void*>> ks(n_layers), vs(n_layers);
    for (int l = 0; l < n_layers; ++l)
      for (int i = 0; i < spans_per_req; ++i) {
        ks[l].push_back(k_spans[((size_t)r * n_layers + l) * spans_per_req + i]);
        vs[l].push_back(v_spans[((size_t)r * n_layers + l) * spans_per_req + i]);
      }
    // the caches already hold `step` tokens (prefill / earlier steps of the test wrote them)
    gc->virtual_k_cache = std::make_shared<ListVirtualCache>(std::move(ks), m->ctx.GetCacheSpanSize(), (size_t)gc->step);
    gc->virtual_v_cache = std::make_shared<ListVirtualCache>(std::move(vs), m->ctx.GetCacheSpanSize(), (size_t)gc->step);
    m->rt.gen_ctx_list.push_back(gc);
  }
  return 0;
}
int dihost_set_prefix_len(dihost_model_t m, int request, int prefix_len) {
  if (request < 0 || request >= (int)m->rt.gen_ctx_list.size()) return (int)AsStatus::ALLSPARK_PARAM_ERROR;
  m->rt.gen_ctx_list[request]->prefix_len = prefix_len;
  return 0;
}
static AsOperator* get_op(dihost_model_t m, int id) { return id >= 0 && id < (int)m->ops.size() ? m->ops[id].get() : nullptr; }
int dihost_op_reshape(dihost_model_t m, int id) {
  AsOperator* op = get_op(m, id);
  return op ? (int)op->CallReshape(&m->rt) : (int)AsStatus::ALLSPARK_PARAM_ERROR;
}
int dihost_op_alloc(dihost_model_t m, int id) {
  AsOperator* op = get_op(m, id);
  return op ? (int)op->CallAlloc(&m->rt) : (int)AsStatus::ALLSPARK_PARAM_ERROR;
}
// CallAlloc of several operators at once, one thread each, as the model does with CONFIG_CONCURRENT_SPAN
// (model.cpp:1253-1262); returns the first non-success status
int dihost_ops_alloc_concurrent(dihost_model_t m, const int* ids, int count) {
  std::vector<int> rc(count, 0);
  std::vector<std::thread> th;
  for (int i = 0; i < count; ++i)
    th.emplace_back([&, i] {
      AsOperator* op = get_op(m, ids[i]);
      rc[i] = op ? (int)op->CallAlloc(&m->rt) : (int)AsStatus::ALLSPARK_PARAM_ERROR;
    });
  for (auto& t : th) t.join();
  for (int v : rc)
    if (v) return v;
  return 0;
}
// cached sequence length of a request's K cache for a layer (VirtualCache::GetSeqLength), -1 if unknown
long dihost_cache_seq_len(dihost_model_t m, int request, int layer) {
  if (request < 0 || request >= (int)m->rt.gen_ctx_list.size() || !m->rt.gen_ctx_list[request]->virtual_k_cache) return -1;
  return (long)m->rt.gen_ctx_list[request]->virtual_k_cache->GetSeqLength(layer);
}

int dihost_op_forward(dihost_model_t m, int id) {
  AsOperator* op = get_op(m, id);
  return op ? (int)op->CallForward(&m->rt) : (int)AsStatus::ALLSPARK_PARAM_ERROR;
}

// ---- the model runner (host/model_runner.h): the reference's operator LIST, optionally rewritten by the fusion pass ----------
int dihost_graph_add_op(dihost_model_t m, const char* op_type, const char* op_name, const char* inputs, const char* outputs,
                        const char* weights, const char* attrs) {
  OperatorProto proto;
  const int rc = parse_proto(proto, op_type, op_name, inputs, outputs, weights, attrs);
  if (rc) return rc;
  m->graph.push_back(std::move(proto));
  return 0;
}
// The operator lists of a SERIALIZED TransformerProto (the reference converter's output, graph_wire.h), appended in the order
// given -- `graphs`: comma-separated graph names, NULL / "" = "decoder,gen_graph", the two AsModel runs per step
// (model.cpp:566-650 / :1248-1325; pre_graph / post_graph are id bookkeeping on the host: the model runner's own).
int dihost_graph_add_serialized(dihost_model_t m, const void* data, size_t bytes, const char* graphs) {
  std::map<std::string, std::vector<OperatorProto>> all;
  std::vector<std::string> names;
  if (!data || !wire::parse_transformer(std::string_view(reinterpret_cast<const char*>(data), bytes), all, names)) {
    g_err = "not a serialized TransformerProto (allspark.proto)";
    return (int)AsStatus::ALLSPARK_PARAM_ERROR;
  }
  for (const std::string& g : split(graphs && graphs[0] ? graphs : "decoder,gen_graph", ',')) {
    auto it = all.find(g);
    if (it == all.end()) {
      g_err = "no graph named " + g + " in the serialized model";
      return (int)AsStatus::ALLSPARK_PARAM_ERROR;
    }
    for (const OperatorProto& op : it->second) m->graph.push_back(op);
  }
  return 0;
}
int dihost_graph_build(dihost_model_t m, int fuse) {
  // ONE build per model: the weight-only operators release the model-owned source tensors once re-laid-out (PackedLowp::Pack,
  // GemmLowpHIP::InitV2), so a second build -- fuse = 0 after fuse = 1, a retry after a failed CallInit -- would pack from freed memory
  if (m->runner) {
    g_err = "graph_build: this model has been built already (its source weights were released after the re-layout): create a new model";
    return (int)AsStatus::ALLSPARK_INVALID_CALL_ERROR;
  }
  m->runner = std::make_unique<HipModelRunner>(&m->ctx, &m->tensors, &m->weights, &m->weights_buffer);
  const AsStatus st = m->runner->Build(m->graph, fuse != 0);
  if (st != AsStatus::ALLSPARK_SUCCESS) g_err = m->runner->last_error();
  return (int)st;
}
// "fused=<0|1>;layers=<n>;ops=<before>-><after>;why=<text>;types=<comma-separated operator types of the built list>"
const char* dihost_graph_report(dihost_model_t m) {
  if (!m->runner) return "";
  const FusionReport& r = m->runner->fusion();
  std::string t = "fused=" + std::to_string(r.fused ? 1 : 0) + ";device_resident=" + std::to_string(r.device_resident ? 1 : 0) + ";layers=" +
                  std::to_string(r.layers) + ";ops=" + std::to_string(r.ops_before) + "->" + std::to_string(r.ops_after) + ";why=" + r.why + ";types=";
  for (size_t i = 0; i < m->runner->protos().size(); ++i) t += (i ? "," : "") + m->runner->protos()[i].op_type;
  m->text = t;
  return m->text.c_str();
}
// the fusion pass alone on the recorded list (no operator is created: runs without a GPU); same report format
const char* dihost_graph_fuse_dry(dihost_model_t m) {
  FusionReport r;
  const std::vector<OperatorProto> out = FuseDecoderGraph(m->graph, m->ctx, &r);
  std::string t = "fused=" + std::to_string(r.fused ? 1 : 0) + ";device_resident=" + std::to_string(r.device_resident ? 1 : 0) + ";layers=" +
                  std::to_string(r.layers) + ";ops=" + std::to_string(r.ops_before) + "->" + std::to_string(r.ops_after) + ";why=" + r.why + ";types=";
  for (size_t i = 0; i < out.size(); ++i) t += (i ? "," : "") + out[i].op_type;
  t += ";wiring=";
  for (size_t i = 0; i < out.size(); ++i) {
    t += (i ? "|" : "") + out[i].op_type + "(";
    for (size_t j = 0; j < out[i].inputs.size(); ++j) t += (j ? "," : "") + out[i].inputs[j];
    t += ")->(";
    for (size_t j = 0; j < out[i].outputs.size(); ++j) t += (j ? "," : "") + out[i].outputs[j];
    t += ")[";
    for (size_t j = 0; j < out[i].weights.size(); ++j) t += (j ? "," : "") + out[i].weights[j];
    t += "]";
  }
  m->text = t;
  return m->text.c_str();
}
static std::shared_ptr<GenerateContext> make_request(dihost_model_t m, int step, int prefix_len, int top_k, float top_p, float temperature,
                                                     unsigned long long seed, int n_layers, int spans_per_req, void* const* k_spans,
                                                     void* const* v_spans, int cached_len) {
  auto gc = std::make_shared<GenerateContext>();
  gc->step = step;
  gc->prefix_len = prefix_len;
  gc->gen_cfg.top_k = top_k;
  gc->gen_cfg.top_p = top_p;
  gc->gen_cfg.temperature = temperature;
  gc->gen_cfg.seed = seed;
  if (m->has_next_gen) {  // dihost_next_request_generation: the rest of the generation config, for this request only
    const GenerateConfig& n = m->next_gen;
    gc->gen_cfg.repetition_penalty = n.repetition_penalty;
    gc->gen_cfg.frequency_penalty = n.frequency_penalty;
    gc->gen_cfg.presence_penalty = n.presence_penalty;
    gc->gen_cfg.no_repeat_ngram_size = n.no_repeat_ngram_size;
    gc->gen_cfg.min_length = n.min_length;
    gc->gen_cfg.eos_token_id = n.eos_token_id;
    gc->gen_cfg.suppress_repetition_in_generation = n.suppress_repetition_in_generation;
    gc->gen_cfg.logprobs = n.logprobs;
    gc->gen_cfg.top_logprobs = n.top_logprobs;
    m->has_next_gen = false;
  }
  std::vector<std::vector<void*>> ks(n_layers), vs(n_layers);
  for (int l = 0; l < n_layers; ++l)
    for (int i = 0; i < spans_per_req; ++i) {
      ks[l].push_back(k_spans[(size_t)l * spans_per_req + i]);
      vs[l].push_back(v_spans[(size_t)l * spans_per_req + i]);
    }
  gc->virtual_k_cache = std::make_shared<ListVirtualCache>(std::move(ks), m->ctx.GetCacheSpanSize(), (size_t)cached_len);
  gc->virtual_v_cache = std::make_shared<ListVirtualCache>(std::move(vs), m->ctx.GetCacheSpanSize(), (size_t)cached_len);
  return gc;
}
static void fill_generation(GenerateConfig& g, float repetition_penalty, float frequency_penalty, float presence_penalty, int no_repeat_ngram_size,
                            int min_length, int eos_token_id, int suppress_repetition_in_generation, int logprobs, int top_logprobs) {
  g.repetition_penalty = repetition_penalty;
  g.frequency_penalty = frequency_penalty;
  g.presence_penalty = presence_penalty;
  g.no_repeat_ngram_size = no_repeat_ngram_size;
  g.min_length = min_length;
  g.eos_token_id = eos_token_id;
  g.suppress_repetition_in_generation = suppress_repetition_in_generation != 0;
  g.logprobs = logprobs != 0;
  g.top_logprobs = top_logprobs;
}
// The part of GenerateConfig (csrc/interface/allspark.h:112-146) GenerateOp's logits processors and log-probability outputs read, for the
// NEXT request started through dihost_request_start (then forgotten) ...
int dihost_next_request_generation(dihost_model_t m, float repetition_penalty, float frequency_penalty, float presence_penalty,
                                   int no_repeat_ngram_size, int min_length, int eos_token_id, int suppress_repetition_in_generation, int logprobs,
                                   int top_logprobs) {
  if (!m) return (int)AsStatus::ALLSPARK_PARAM_ERROR;
  fill_generation(m->next_gen, repetition_penalty, frequency_penalty, presence_penalty, no_repeat_ngram_size, min_length, eos_token_id,
                  suppress_repetition_in_generation, logprobs, top_logprobs);
  m->has_next_gen = true;
  return 0;
}
// ... and for request `index` of the runtime context set by dihost_set_runtime (operator-by-operator use); input_len: its prompt length
int dihost_request_generation(dihost_model_t m, int index, float repetition_penalty, float frequency_penalty, float presence_penalty,
                              int no_repeat_ngram_size, int min_length, int eos_token_id, int suppress_repetition_in_generation, int logprobs,
                              int top_logprobs, int input_len) {
  if (!m || index < 0 || index >= (int)m->rt.gen_ctx_list.size()) return (int)AsStatus::ALLSPARK_PARAM_ERROR;
  fill_generation(m->rt.gen_ctx_list[index]->gen_cfg, repetition_penalty, frequency_penalty, presence_penalty, no_repeat_ngram_size, min_length,
                  eos_token_id, suppress_repetition_in_generation, logprobs, top_logprobs);
  m->rt.gen_ctx_list[index]->input_len = input_len;
  return 0;
}
// Log-probabilities of a request: `count` records starting at record `first` -> token_logprob [count], top_value / top_index [count, top_n]
// (top_n <= 10).  runner != 0: request `index` of the model runner's running batch, records indexed by the token's POSITION in the sequence
// (the first generated token of an L-token prompt is record L), read from the request's device-resident log after a stream synchronise;
// runner == 0: request `index` of the runtime context, records in generation order (Request::log_probs_list / token_logprobs_list).
// -> records copied, or a negative AsStatus.
int dihost_request_logprobs(dihost_model_t m, int index, int runner, int first, int count, int top_n, float* token_logprob, float* top_value,
                            int* top_index) {
  if (!m || first < 0 || count < 0 || top_n < 0 || top_n > 10) return -(int)AsStatus::ALLSPARK_PARAM_ERROR;
  const int S = m->ctx.GetMaxTopLogprobs();
  if (runner) {
    if (!m->runner || index < 0 || index >= (int)m->runner->running().size()) return -(int)AsStatus::ALLSPARK_PARAM_ERROR;
    auto& gc = m->runner->running()[index];
    if (!gc->logprob_records_dev) return -(int)AsStatus::ALLSPARK_INVALID_CALL_ERROR;
    const int words = 1 + 2 * S;
    const int64_t have = gc->logprob_records_dev->Count() / words;
    const int n = (int)std::max<int64_t>(0, std::min<int64_t>(count, have - first));
    std::vector<float> rec((size_t)n * words);
    if (hipStreamSynchronize(m->ctx.GetStream()) != hipSuccess ||
        (n > 0 && hipMemcpy(rec.data(), (const float*)gc->logprob_records_dev->GetDataPtr() + (size_t)first * words, rec.size() * sizeof(float),
                            hipMemcpyDeviceToHost) != hipSuccess))
      return -(int)AsStatus::ALLSPARK_RUNTIME_ERROR;
    for (int i = 0; i < n; ++i) {
      const float* r = rec.data() + (size_t)i * words;
      if (token_logprob) token_logprob[i] = r[0];
      for (int k = 0; k < top_n; ++k) {
        if (top_value) top_value[(size_t)i * top_n + k] = r[1 + k];
        if (top_index) std::memcpy(&top_index[(size_t)i * top_n + k], &r[1 + S + k], sizeof(int));
      }
    }
    return n;
  }
  if (index < 0 || index >= (int)m->rt.gen_ctx_list.size() || !m->rt.gen_ctx_list[index]->request) return -(int)AsStatus::ALLSPARK_PARAM_ERROR;
  Request& rq = *m->rt.gen_ctx_list[index]->request;
  std::lock_guard<std::mutex> g(rq.queue_mu);
  const int n = std::max(0, std::min(count, (int)rq.token_logprobs_list.size() - first));
  for (int i = 0; i < n; ++i) {
    if (token_logprob) token_logprob[i] = rq.token_logprobs_list[first + i];
    const auto& top = rq.log_probs_list[first + i];
    for (int k = 0; k < top_n; ++k) {
      if (top_value) top_value[(size_t)i * top_n + k] = k < (int)top.size() ? top[k].second : -std::numeric_limits<float>::infinity();
      if (top_index) top_index[(size_t)i * top_n + k] = k < (int)top.size() ? top[k].first : -1;
    }
  }
  return n;
}
int dihost_request_start(dihost_model_t m, const int64_t* prompt_host, int len, int prefix_len, int top_k, float top_p, float temperature,
                         unsigned long long seed, int n_layers, int spans_per_req, void* const* k_spans, void* const* v_spans,
                         int64_t* first_id) {
  if (!m->runner) return (int)AsStatus::ALLSPARK_INVALID_CALL_ERROR;
  auto gc = make_request(m, prefix_len, prefix_len, top_k, top_p, temperature, seed, n_layers, spans_per_req, k_spans, v_spans, prefix_len);
  const AsStatus st = m->runner->StartRequest(gc, prompt_host, len, first_id);
  if (st != AsStatus::ALLSPARK_SUCCESS) g_err = m->runner->last_error();
  return (int)st;
}
int dihost_request_adopt(dihost_model_t m, int cached_len, int64_t next_id, int n_layers, int spans_per_req, void* const* k_spans,
                         void* const* v_spans) {
  if (!m->runner) return (int)AsStatus::ALLSPARK_INVALID_CALL_ERROR;
  auto gc = make_request(m, cached_len, 0, 1, 1.0f, 1.0f, 0, n_layers, spans_per_req, k_spans, v_spans, cached_len);
  const AsStatus st = m->runner->AdoptRequest(gc, next_id);
  if (st != AsStatus::ALLSPARK_SUCCESS) g_err = m->runner->last_error();
  return (int)st;
}
int dihost_request_stop(dihost_model_t m, int index) {
  if (!m->runner) return (int)AsStatus::ALLSPARK_INVALID_CALL_ERROR;
  const AsStatus st = m->runner->StopRequest(index);
  if (st != AsStatus::ALLSPARK_SUCCESS) g_err = m->runner->last_error();
  return (int)st;
}
int dihost_decode_steps(dihost_model_t m, int n, int use_graph) {
  if (!m->runner) return (int)AsStatus::ALLSPARK_INVALID_CALL_ERROR;
  const AsStatus st = m->runner->DecodeSteps(n, use_graph != 0);
  if (st != AsStatus::ALLSPARK_SUCCESS) g_err = m->runner->last_error();
  return (int)st;
}
int dihost_sync_ids(dihost_model_t m, int64_t* ids_host, int capacity) {
  if (!m->runner) return -1;
  std::vector<int64_t> ids;
  const AsStatus st = m->runner->Sync(&ids);
  if (st != AsStatus::ALLSPARK_SUCCESS) {
    g_err = m->runner->last_error();
    if (m->runner->handoff_failed()) {  // the harness owns the caches: back to the rolled-back lengths (an engine would fail the requests)
      for (auto& gc : m->runner->running()) {
        auto* k = dynamic_cast<ListVirtualCache*>(gc->virtual_k_cache.get());
        auto* v = dynamic_cast<ListVirtualCache*>(gc->virtual_v_cache.get());
        if (k && v) {
          k->Rewind((size_t)gc->step);
          v->Rewind((size_t)gc->step);
        }
      }
    }
    return -(int)st;
  }
  for (int i = 0; i < (int)ids.size() && i < capacity; ++i) ids_host[i] = ids[i];
  return (int)ids.size();
}
// ---- the Request slice of a runtime-context request (id-processing operators: PreProcessId / UpdateId / PostProcessId) -------
// attaches a Request with inputs["input_ids"] = ids [1, len] (host) to request `index` of the runtime context set by
// dihost_set_runtime, with the stop conditions UpdateId checks; stop_words: `n_words` sequences of `word_len` ids each
int dihost_request_attach(dihost_model_t m, int index, const int64_t* ids, int len, int max_length, int early_stopping, int eos_token_id,
                          const int64_t* stop_words, int n_words, int word_len, int in_length_bias) {
  if (index < 0 || index >= (int)m->rt.gen_ctx_list.size() || !ids || len <= 0) return (int)AsStatus::ALLSPARK_PARAM_ERROR;
  auto& gc = m->rt.gen_ctx_list[index];
  auto rq = std::make_shared<Request>();
  rq->request_id = "request-" + std::to_string(index);
  auto t = std::make_shared<AsTensor>("input_ids", DeviceType::CPU, INT64, Shape{1, len});
  std::memcpy(t->GetDataPtr(), ids, (size_t)len * sizeof(int64_t));
  rq->inputs["input_ids"] = t;
  gc->request = rq;
  gc->gen_cfg.max_length = max_length;
  gc->gen_cfg.early_stopping = early_stopping != 0;
  gc->gen_cfg.eos_token_id = eos_token_id;
  gc->gen_cfg.stop_words_ids.clear();
  for (int w = 0; w < n_words; ++w) gc->gen_cfg.stop_words_ids.emplace_back(stop_words + (size_t)w * word_len, stop_words + (size_t)(w + 1) * word_len);
  gc->in_length_bias = in_length_bias;
  gc->finish = false;
  gc->gen_over[0] = false;
  gc->engine_max_length = m->ctx.GetModelMaxLength();
  return 0;
}
// what GenerateOp's fill_generated_ids does after sampling (generate_impl_cpu.hpp:176-200): token -> generated_ids[position], shape grown
int dihost_request_put_token(dihost_model_t m, int index, int position, int64_t token) {
  if (index < 0 || index >= (int)m->rt.gen_ctx_list.size() || !m->rt.gen_ctx_list[index]->request) return (int)AsStatus::ALLSPARK_PARAM_ERROR;
  auto& interim = m->rt.gen_ctx_list[index]->request->interim;
  auto it = interim.find("generated_ids");
  if (it == interim.end()) return (int)AsStatus::ALLSPARK_INVALID_CALL_ERROR;
  AsTensor& t = *it->second;
  if (position < 0) return (int)AsStatus::ALLSPARK_PARAM_ERROR;
  if (position >= t.Count() && t.SetShape(Shape{1, (int64_t)position + 1}) != AsStatus::ALLSPARK_SUCCESS) return (int)AsStatus::ALLSPARK_EXCEED_LIMIT_ERROR;
  static_cast<int64_t*>(t.GetDataPtr())[position] = token;
  return 0;
}
int dihost_set_phase(dihost_model_t m, int is_context) {  // the same requests, the other phase (context <-> decoder)
  m->rt.is_context = is_context != 0;
  return 0;
}
int dihost_request_set_step(dihost_model_t m, int index, int step, int in_length_bias) {
  if (index < 0 || index >= (int)m->rt.gen_ctx_list.size()) return (int)AsStatus::ALLSPARK_PARAM_ERROR;
  m->rt.gen_ctx_list[index]->step = step;
  m->rt.gen_ctx_list[index]->in_length_bias = in_length_bias;
  return 0;
}
// -> number of queued tokens copied (the queue is drained), *finish = the request's stop verdict; negative AsStatus on error
int dihost_request_poll(dihost_model_t m, int index, int64_t* tokens, int capacity, int* finish, int* n_interim) {
  if (index < 0 || index >= (int)m->rt.gen_ctx_list.size() || !m->rt.gen_ctx_list[index]->request) return -(int)AsStatus::ALLSPARK_PARAM_ERROR;
  auto& gc = m->rt.gen_ctx_list[index];
  std::lock_guard<std::mutex> g(gc->request->queue_mu);
  auto& q = gc->request->generated_ids_queue;
  const int n = std::min((int)q.size(), capacity);
  for (int i = 0; i < n; ++i) tokens[i] = q[i];
  q.erase(q.begin(), q.begin() + n);
  if (finish) *finish = gc->finish ? 1 : 0;
  if (n_interim) *n_interim = (int)gc->request->interim.size();
  return n;
}
int dihost_running_batch(dihost_model_t m) { return m->runner ? m->runner->batch() : 0; }
int dihost_requests_rewind(dihost_model_t m, int cached_len) {
  if (!m->runner) return (int)AsStatus::ALLSPARK_INVALID_CALL_ERROR;
  for (auto& gc : m->runner->running()) {
    auto* k = dynamic_cast<ListVirtualCache*>(gc->virtual_k_cache.get());
    auto* v = dynamic_cast<ListVirtualCache*>(gc->virtual_v_cache.get());
    if (!k || !v) return (int)AsStatus::ALLSPARK_PARAM_ERROR;
    k->Rewind((size_t)cached_len);
    v->Rewind((size_t)cached_len);
  }
  const AsStatus st = m->runner->Rewind(cached_len);
  if (st != AsStatus::ALLSPARK_SUCCESS) g_err = m->runner->last_error();
  return (int)st;
}

}  // extern "C"
