// as_types.h -- the slice of the allspark core types the hot-path operators touch, restated so that
// the HIP operator layer compiles stand-alone (the reference's operator.h pulls in <dnnl.hpp>,
// protobuf and glog, none of which exist in this image -- SURVEY F4).  Names, member functions
// and semantics follow the reference so that the operator sources below move into
// csrc/core/operator/ unchanged once DeviceType::HIP exists there:
//   AsStatus        csrc/interface/allspark_check.h:62-80
//   DataType/Device csrc/interface/allspark.h, csrc/proto/allspark.proto
//   AsTensor        csrc/core/tensor/tensor.h:54-160   (name, dtype, shape, data pointer, Free/SetShape)
//   TensorMap       name -> shared_ptr<AsTensor>
//   DeviceContext   csrc/device/device_context.h:17-209 (model dims, cache mode, span size, rank info)
//   RuntimeContext / GenerateContext  csrc/core/model/generate_context.h:32-70,95+
//   OperatorProto   csrc/proto/allspark.proto (op_type, op_name, inputs, outputs, weights, attr map of
//                   raw little-endian bytes)
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace allspark {

enum class AsStatus : int {
  ALLSPARK_SUCCESS = 0,
  ALLSPARK_UNKNOWN_ERROR = 1,
  ALLSPARK_PARAM_ERROR = 2,
  ALLSPARK_IO_ERROR = 3,
  ALLSPARK_MEMORY_ERROR = 4,
  ALLSPARK_RUNTIME_ERROR = 5,
  ALLSPARK_EXCEED_LIMIT_ERROR = 7,
  ALLSPARK_INVALID_CALL_ERROR = 8,
  ALLSPARK_CACHE_MEMORY_OUT = 11,
};
#define AS_CHECK_STATUS(expr)                                     \
  do {                                                            \
    ::allspark::AsStatus s_ = (expr);                             \
    if (s_ != ::allspark::AsStatus::ALLSPARK_SUCCESS) return s_;  \
  } while (0)

enum DataType { DATATYPE_UNDEFINED = 0, FLOAT32 = 1, FLOAT16 = 2, INT8 = 3, INT16 = 4, INT32 = 5, INT64 = 6,
                STRING = 7, BOOL = 8, BFLOAT16 = 9, UINT8 = 10, POINTER = 20 };
enum DeviceType { DEVICETYPE_UNDEFINED = 0, CPU = 1, CUDA = 2, COMPILE_TIME_MAX_DEVICE = 3, HIP = 4 };
enum UnaryType { UNARYTYPE_UNDEFINED = 0, TANH = 1, GELU_ERF = 2, GELU_TANH = 3, RELU = 4, SILU = 5, SIGMOID = 6 };
enum class AsCacheMode { AsCacheDefault = 0, AsCacheQuantI8 = 1, AsCacheQuantU4 = 2 };

inline size_t SizeofType(DataType t) {
  switch (t) {
    case FLOAT32: case INT32: return 4;
    case FLOAT16: case BFLOAT16: case INT16: return 2;
    case INT64: case POINTER: return 8;
    default: return 1;
  }
}

using Shape = std::vector<int64_t>;

// Device tensor: owns its storage unless constructed as a view.
class AsTensor {
 public:
  AsTensor(const std::string& name, DeviceType dev, DataType dtype, const Shape& shape = {})
      : name_(name), dev_(dev), dtype_(dtype) {
    SetShape(Shape(shape));
  }
  AsTensor(const std::string& name, DeviceType dev, DataType dtype, const Shape& shape, void* view)
      : name_(name), dev_(dev), dtype_(dtype), shape_(shape), data_(view), owner_(false) {}
  ~AsTensor() { Free(); }
  AsTensor(const AsTensor&) = delete;
  AsTensor& operator=(const AsTensor&) = delete;

  const std::string& GetName() const { return name_; }
  DataType GetDataType() const { return dtype_; }
  void SetDataType(DataType t) { dtype_ = t; }
  DeviceType GetDeviceType() const { return dev_; }
  const Shape& GetShape() const { return shape_; }
  int64_t Count() const {
    int64_t c = 1;
    for (auto d : shape_) c *= d;
    return shape_.empty() ? 0 : c;
  }
  size_t GetSizeInByte() const { return (size_t)Count() * SizeofType(dtype_); }
  void* GetDataPtr() const { return data_; }
  // grows only (like the reference's block allocator behind SetShape)
  AsStatus SetShape(Shape&& shape) {
    shape_ = std::move(shape);
    const size_t need = GetSizeInByte();
    if (!owner_) return need <= capacity_ || capacity_ == 0 ? AsStatus::ALLSPARK_SUCCESS : AsStatus::ALLSPARK_MEMORY_ERROR;
    if (need > capacity_) {
      Free();
      if (dev_ == CPU) {
        data_ = malloc(need);
      } else if (hipMalloc(&data_, need) != hipSuccess) {
        (void)hipGetLastError();
        data_ = nullptr;
        return AsStatus::ALLSPARK_MEMORY_ERROR;
      }
      capacity_ = need;
    }
    return AsStatus::ALLSPARK_SUCCESS;
  }
  bool OwnsStorage() const { return owner_; }
  void Free() {
    if (owner_ && data_) {
      if (dev_ == CPU) free(data_);
      else (void)hipFree(data_);
    }
    data_ = nullptr;
    capacity_ = 0;
  }

 private:
  std::string name_;
  DeviceType dev_;
  DataType dtype_;
  Shape shape_;
  void* data_ = nullptr;
  size_t capacity_ = 0;
  bool owner_ = true;
};
using TensorMap = std::map<std::string, std::shared_ptr<AsTensor>>;

struct OperatorProto {
  std::string op_type, op_name;
  std::vector<std::string> inputs, outputs, weights;
  std::map<std::string, std::string> attr;  // raw bytes, read as *(T*)attr.at(k).c_str() like the reference
};

// csrc/device/device_context.h + the HIP specifics (stream, RCCL communicator)
class DeviceContext {
 public:
  virtual ~DeviceContext() = default;
  virtual DeviceType GetDeviceType() const = 0;
  int GetNumberHeads() const { return num_heads_; }
  int GetNumberGroups() const { return num_groups_; }
  int GetSizePerHead() const { return size_per_head_; }
  int GetCacheSpanSize() const { return span_size_; }
  AsCacheMode GetCacheMode() const { return cache_mode_; }
  int GetModelMaxBatch() const { return max_batch_; }
  int GetModelMaxLength() const { return max_length_; }
  int GetMaxTopLogprobs() const { return 10; }  // device_context.h:135,182 (engine_max_top_logprobs, const)
  int GetRank() const { return rank_; }
  int GetNranks() const { return nranks_; }
  void SetNumberHeads(int v) { num_heads_ = v; }
  void SetNumberGroups(int v) { num_groups_ = v; }
  void SetSizePerHead(int v) { size_per_head_ = v; }
  void SetCacheSpanSize(int v) { span_size_ = v; }
  void SetCacheMode(AsCacheMode m) { cache_mode_ = m; }
  void SetModelMaxBatch(int v) { max_batch_ = v; }
  void SetModelMaxLength(int v) { max_length_ = v; }
  void SetRankInfo(int rank, int nranks) { rank_ = rank; nranks_ = nranks; }

 protected:
  int num_heads_ = 0, num_groups_ = 0, size_per_head_ = 0, span_size_ = 0, max_batch_ = 1, max_length_ = 0;
  int rank_ = 0, nranks_ = 1;
  AsCacheMode cache_mode_ = AsCacheMode::AsCacheDefault;
};

// A consumer GEMM's shape, advertised at Init for the tensor it reads, so that the PRODUCER of that tensor can decide at
// Reshape (when the row count is known) whether to write it in the MFMA-fragment layout the small-batch kernels load
// fastest (DIHIP_ACT_FRAG32, include/dashinfer_hip.h section 1) -- the decision decoder.DecodeSession takes with
// ops.prefers_frag().  Backend-private graph annotation: nothing of it crosses the operator interface.
struct ActLayoutPref {
  int wbits = 0, n = 0, k = 0, group = -1, dual = 0, bf16 = 1;
};
// Deferred RMSNorm (include/dashinfer_hip.h, "with the RMSNorm DEFERRED"): what the producer of a pre-normalised tensor left for
// its consumer in this step -- parts > 0: the tensor holds FT(gamma * h), `rowsq` the partial sums of h^2, and the consumer scales
// its accumulators by 1 / rms; parts == 0: the tensor is the finished norm.  Written at the producer's Forward, read at the
// consumer's Forward of the same step (list order).
struct RowNormState {
  const float* rowsq = nullptr;
  int parts = 0;
  float eps = 0.f;
};

class HIPContext : public DeviceContext {
 public:
  DeviceType GetDeviceType() const override { return DeviceType::HIP; }
  hipStream_t GetStream() const { return stream_; }
  void SetStream(hipStream_t s) { stream_ = s; }
  void* GetRCCLComm() const { return comm_; }
  void SetRCCLComm(void* c) { comm_ = c; }
  // the one-shot peer-to-peer communicator for decode-sized messages (dihip_p2p_ar_create, include/dashinfer_hip.h section 6b), or
  // null: the AllReduce operator uses it for messages up to dihip_p2p_ar_max_bytes() and RCCL beyond (bench.py --gpus N does the
  // same choice in the Python runner); it is also what lets rank THREADS of one process on one GPU stand in for a node in tests
  void* GetP2PComm() const { return p2p_comm_; }
  void SetP2PComm(void* c) { p2p_comm_ = c; }
  // The sequence length the decode-step LAUNCH PLANS are made for (split count and split width of the paged attention, the fused attention
  // block's grid): the running requests' length rounded up to a bucket, never the engine's maximum length -- with an 8192-token engine a
  // 2048-token request would otherwise be split 64 ways (47 empty splits to merge) and the attention block, which serves up to 28 splits,
  // would never run.  The model runner moves it (host/model_runner.cpp DecodeSteps: Reshape + a new captured step when the bucket changes);
  // buffers keep being sized by GetModelMaxLength().  0 / unset: the maximum length.
  int PlanLength() const { return plan_len_ > 0 ? std::min(plan_len_, max_length_) : max_length_; }
  void SetPlanLength(int n) { plan_len_ = n; }

  // ---- annotations of the fused decode graph (host/fused_ops_hip.cpp; written at Init / Reshape, read at Forward) ----
  void AdvertiseLayoutPref(const std::string& tensor, const ActLayoutPref& p) const { layout_pref_[tensor] = p; }
  const ActLayoutPref* LayoutPref(const std::string& tensor) const {
    auto it = layout_pref_.find(tensor);
    return it == layout_pref_.end() ? nullptr : &it->second;
  }
  void SetActLayout(const std::string& tensor, int layout) const { act_layout_[tensor] = layout; }
  int ActLayout(const std::string& tensor) const {
    auto it = act_layout_.find(tensor);
    return it == act_layout_.end() ? 0 : it->second;
  }
  // The model runner keeps the per-request sequence lengths ON THE DEVICE (tensors "dihip.old_seq_lens" / "dihip.new_seq_lens",
  // advanced by the greedy sampling launch) so that a captured decode step replays with nothing changing on the host; without a
  // runner (a plain Alloc -> Forward loop over the operators) every attention operator uploads the lengths itself per step.
  bool LensOnDevice() const { return lens_on_device_; }
  void SetLensOnDevice(bool v) const { lens_on_device_ = v; }
  // Which operator produces a tensor of the fused list (registered at Init, in list order): the o-projection finds the attention
  // and qkv operators in front of it and, for one request on the 16-bit cache, runs all three as ONE launch
  // (dihip_decode_attn_block; host/fused_ops_hip.cpp DihipGemmAddTo).  Opaque here: the operators cast to their own interfaces.
  void SetRowNorm(const std::string& tensor, const RowNormState& r) const { row_norm_[tensor] = r; }
  RowNormState RowNorm(const std::string& tensor) const {
    auto it = row_norm_.find(tensor);
    return it == row_norm_.end() ? RowNormState{} : it->second;
  }
  // the fused attention block's hand-offs timed out once (model_runner.cpp Sync read the error word): the launch chain serves from then on
  // Tensor-parallel all-reduce beside the next GEMV's weights (north star: "overlapped with the next GEMM on a side HIP stream"): the
  // weight-streaming operators register their packed weights under the name of the hidden-row tensor they READ; the AllReduce operator
  // that WRITES that tensor runs its collective on the side stream between two events while the main stream pulls those weights on-die
  // (dihip_prefetch), and joins.  DIHIP_TP_OVERLAP=1 (off by default: a fork / join inside a captured step measured ~20 us on this
  // runtime; the reference synchronises the host instead, allreduce_op.cpp:84-92).
  struct WeightSpan {
    const void* ptr;
    size_t bytes;
  };
  void RegisterConsumerWeights(const std::string& tensor, const void* ptr, size_t bytes) const {
    if (ptr && bytes) consumer_w_[tensor].push_back(WeightSpan{ptr, bytes});
  }
  const std::vector<WeightSpan>* ConsumerWeights(const std::string& tensor) const {
    auto it = consumer_w_.find(tensor);
    return it == consumer_w_.end() ? nullptr : &it->second;
  }
  hipStream_t SideStream() const {  // created on first use, lives as long as the context
    if (!side_ && hipStreamCreateWithFlags(&side_, hipStreamNonBlocking) != hipSuccess) side_ = nullptr;
    return side_;
  }
  // the tensor the graph's first operator reads the token ids from (the model runner sets it; the sampling operator reads the prompt
  // length off its shape in the context phase)
  const std::string& InputIdsName() const { return ids_name_; }
  void SetInputIdsName(const std::string& n) const { ids_name_ = n; }
  bool AttnBlockDisabled() const { return attn_block_off_; }
  void DisableAttnBlock() const { attn_block_off_ = true; }
  void RegisterProducer(const std::string& tensor, void* op) const { producer_[tensor] = op; }
  void* Producer(const std::string& tensor) const {
    auto it = producer_.find(tensor);
    return it == producer_.end() ? nullptr : it->second;
  }

 private:
  hipStream_t stream_ = nullptr;
  void* comm_ = nullptr;
  void* p2p_comm_ = nullptr;
  int plan_len_ = 0;
  mutable std::map<std::string, ActLayoutPref> layout_pref_;
  mutable std::map<std::string, int> act_layout_;
  mutable bool lens_on_device_ = false;
  mutable bool attn_block_off_ = false;
  mutable std::string ids_name_ = "input_ids";
  mutable std::map<std::string, std::vector<WeightSpan>> consumer_w_;
  mutable hipStream_t side_ = nullptr;
  mutable std::map<std::string, void*> producer_;
  mutable std::map<std::string, RowNormState> row_norm_;
};

// VirtualCache (csrc/runtime/cache/virtual_cache.h:93-139): the per-request paged cache of all layers, as the span
// operators see it.  GetCache(layer, increment) grows the layer's sequence by `increment` tokens -- claiming spans from the
// cache manager when a span boundary is crossed -- and returns the layer's span-pointer vector as a POINTER tensor on the
// host.  The managers behind it (CacheSpanManager, CacheFrameManager, prefix cache) are host code of the reference and
// are reused as they are; only this interface crosses into the operator.
class VirtualCache {
 public:
  virtual ~VirtualCache() = default;
  virtual const AsTensor& GetCache(int layer_id, int increment) = 0;  // throws AsException (PARAM_ERROR, CACHE_MEMORY_OUT)
  virtual size_t GetSeqLength(int layer_id) const = 0;
  virtual int GetLayerNum() const = 0;
};

// the sampling half of GenerateConfig (csrc/interface/allspark.h GenerateConfig: top_k / top_p / temperature / seed, the
// fields GenerateOp reads per request, generate_op.cpp:60-140)
struct GenerateConfig {
  int top_k = 1;            // 1: greedy; 1 .. 1024 served; 0 (the whole vocabulary) and > 1024 are refused (sampling_host.h)
  float top_p = 1.0f;       // 0 or >= 1: off
  float temperature = 1.0f;
  unsigned long long seed = 0;
  // the stop conditions UpdateIdOp checks (update_id_op.cpp:42-75)
  int max_length = 0;       // 0: unbounded here (the engine always sets it)
  bool early_stopping = true;
  int eos_token_id = -1;
  std::vector<std::vector<int64_t>> stop_words_ids;
  // the logits processors GenerateOp runs before sampling and its log-probability outputs (csrc/interface/allspark.h:122-146, defaults alike)
  float repetition_penalty = 1.0f;
  float presence_penalty = 0.f;
  float frequency_penalty = 0.f;
  bool suppress_repetition_in_generation = false;
  int no_repeat_ngram_size = 0;
  int min_length = 0;
  bool logprobs = false;
  int top_logprobs = 0;     // <= 10 (engine_max_top_logprobs, csrc/common/as_engine.h:232)
  // anything but the neutral values: the request needs its token history on the device
  bool has_logits_processors() const {
    return repetition_penalty != 1.0f || presence_penalty != 0.f || frequency_penalty != 0.f || no_repeat_ngram_size != 0 || min_length > 0;
  }
};

// csrc/common/request.h:25-40, the slice the id-processing operators touch: the request's input tensors, its intermediate
// tensors ("generated_ids" on the host, "generated_ids_gpu" / "new_input_ids_gpu" on the device) and the queue of generated
// tokens the engine drains (moodycamel::ConcurrentQueue in the reference; single producer / single consumer here)
struct Request {
  std::string request_id;
  std::map<std::string, std::shared_ptr<class AsTensor>> inputs, interim;
  std::vector<int64_t> generated_ids_queue;
  // request.h:37-39: per generated token, the top_logprobs (token, log-probability) pairs and the chosen token's log-probability
  std::vector<std::vector<std::pair<int, float>>> log_probs_list;
  std::vector<float> token_logprobs_list;
  std::mutex queue_mu;
  bool finish = false;
  void enqueue(int64_t t) {
    std::lock_guard<std::mutex> g(queue_mu);
    generated_ids_queue.push_back(t);
  }
};

// per-request generation state (generate_context.h:32-70): step = tokens already in the cache
struct GenerateContext {
  int step = 0;
  int prefix_len = 0;
  GenerateConfig gen_cfg;
  unsigned long long sample_calls = 0;   // draws taken so far (the per-request random stream's position)
  // generate_context.h:32-55: what UpdateIdOp / PreProcessIdOp read
  int in_length_bias = 0;
  int input_len = 0;                     // generate_context.h:45: prompt length (the logits processors' "generated" tokens start here)
  // model-runner path (device-resident step state): the request's token history [max_length] INT64 and its log-probability records
  // [max_length][1 + 2 * 10] on the device, written by the step itself (csrc/logits_proc.hip: the rows / records forms)
  std::shared_ptr<class AsTensor> history_dev, logprob_records_dev;
  bool finish = false;
  int generate_method = 0;               // sample = 0 (beam search = 1 is refused, generate_op.cpp:655-660)
  bool gen_over[1] = {false};
  int engine_max_length = 0;
  std::shared_ptr<Request> request;
  std::shared_ptr<VirtualCache> virtual_k_cache, virtual_v_cache;  // generate_context.h:60-61
};

struct RuntimeContext {
  bool is_context = false;
  int current_batch = 0;  // prefill: index of the request being prefilled
  std::vector<std::shared_ptr<GenerateContext>> gen_ctx_list;
  int GetGenCtxListSize() const { return (int)gen_ctx_list.size(); }
  GenerateContext* GetGenCtx(int i) const { return gen_ctx_list.at(i).get(); }
  GenerateContext* GetContextGenCtx() const { return gen_ctx_list.at(current_batch).get(); }
};

}  // namespace allspark
