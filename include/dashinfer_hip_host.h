/* dashinfer_hip_host.h -- C test harness around the C++ operator layer (dash-infer_amd/host).
 *
 * NOT part of the drop-in boundary (that is include/dashinfer_hip.h below the operators and the
 * allspark::AsOperator interface above them): this header only lets tests, written in Python,
 * build an OperatorProto / TensorMap / RuntimeContext, look an op up in the OpFactory and drive
 * CallInit / CallReshape / CallAlloc / CallForward the way AsModel does
 * (csrc/core/model/model.cpp:265-287,566-650,1305-1325).  Status codes are AsStatus values.     */
#ifndef DASHINFER_HIP_HOST_H_
#define DASHINFER_HIP_HOST_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct dihost_model* dihost_model_t; /* HIPContext + tensor map + weights map + runtime context */

/* dtype codes are allspark DataType values (FLOAT32 1, FLOAT16 2, INT8 3, INT32 5, BFLOAT16 9, UINT8 10) */
int dihost_model_create(dihost_model_t* m, void* stream, int num_heads, int num_groups, int size_per_head, int span_size,
                        int cache_mode, int max_batch, int max_length, int rank, int nranks, void* rccl_comm);
/* tensor parallelism: the one-shot peer-to-peer communicator of this rank (dihip_p2p_ar_create), used by the AllReduce operator for
 * decode-sized messages; RCCL (rccl_comm above) beyond dihip_p2p_ar_max_bytes().  One model per rank: a process per GPU on a node,
 * or -- tests -- a thread per rank on one GPU (as_engine.cpp:243-286 runs a thread per rank too) */
int dihost_model_set_p2p_comm(dihost_model_t m, void* p2p_comm);
int dihost_model_destroy(dihost_model_t m);
/* tensors are views of caller-owned device memory (torch tensors in the tests) */
int dihost_set_tensor(dihost_model_t m, const char* name, int dtype, int ndim, const int64_t* shape, void* data);
int dihost_set_weight(dihost_model_t m, const char* name, int dtype, int ndim, const int64_t* shape, void* data);
/* A serialized weight file of the reference's converter (".asparam": python/pyhie/allspark/model/model_base.py -> csrc/utility/
 * allsparkz_util.cpp:264-339; reader on the reference side: WeightFileParser, csrc/runtime/weight/weight_loader.cpp) --
 *   _index: its records as text, one per line "name|dtype|d0,d1,...|split_mode|offset|nbytes" (DataType codes of allspark.proto; no
 *           device work, *need = bytes incl. the terminator, `out` may be null);
 *   _load_file: every record becomes a weight of the model under its own name in device memory the MODEL owns (freed by the
 *           weight-only operators once re-laid-out); dense records only, little endian; *count = records loaded.  With nranks > 1 (the
 *           model's rank / nranks of dihost_model_create) every record is split FOR THIS RANK by its SplitMode and group_list on the way in --
 *           WeightManager -> WeightSplitter of the reference (csrc/runtime/weight/weight_splitter.cpp:60-127 VSPLIT, :369-438 HSPLIT incl. the
 *           rank-0-only bias, :611-721 GROUP_VSPLIT, :128-232 / :439-520 BATCH_V/HSPLIT, :521-610 QKV/KVSPLIT, :722-852 MQA_VSPLIT, :853-919
 *           EPSPLIT; host/weight_file.h SliceForRank): ONE export of the whole model feeds any tensor-parallel degree that divides it;
 *   _slice: the same share of ONE record on the host (no GPU, no model; data == NULL: *nbytes / shape only) -- what _load_file uploads for
 *           (rank, nranks);
 *   dihost_get_weight: a weight's type / shape / device pointer (tests); ALLSPARK_INVALID_CALL_ERROR (type and shape filled, *data NULL) for a weight
 *           whose operator released the source after re-laying it out. */
int dihost_weight_file_index(const char* path, char* out, size_t cap, size_t* need);
int dihost_weights_load_file(dihost_model_t m, const char* path, int* count);
int dihost_weight_file_slice(const char* path, const char* name, int rank, int nranks, void* data, size_t capacity, size_t* nbytes,
                             int64_t* shape8, int* ndim);
int dihost_get_weight(dihost_model_t m, const char* name, int* dtype, int* ndim, int64_t* shape8, void** data);
/* output tensors are owned by the model: shape / pointer after Reshape */
int dihost_get_tensor(dihost_model_t m, const char* name, int* dtype, int* ndim, int64_t* shape8, void** data);

/* names are comma-separated; attrs: "key=i:<int>" | "key=f:<float>" | "key=b:<0|1>", ';'-separated.
 * Looks {op_type, HIP} up in the OpFactory; an unregistered type returns ALLSPARK_PARAM_ERROR with
 * dihost_last_error() == "Unsupported op type." */
int dihost_op_create(dihost_model_t m, int* op_id, const char* op_type, const char* op_name, const char* inputs,
                     const char* outputs, const char* weights, const char* attrs);
/* runtime context: is_context, one entry per request: step (tokens in cache); span pointer tables
 * k_spans / v_spans: host arrays [n_requests][n_layers][spans_per_req] of device pointers -- the spans each
 * request's VirtualCache (a list-backed stand-in for SpannedVirtualCache) claims in order as CallAlloc grows it;
 * the caches start at `step` tokens */
int dihost_set_runtime(dihost_model_t m, int is_context, int n_requests, const int* steps, int n_layers, int spans_per_req,
                       void* const* k_spans, void* const* v_spans);
/* prefix-cache hit of the request being prefilled: tokens already present in its first spans */
int dihost_set_prefix_len(dihost_model_t m, int request, int prefix_len);
int dihost_op_reshape(dihost_model_t m, int op_id);
int dihost_op_alloc(dihost_model_t m, int op_id);
int dihost_op_forward(dihost_model_t m, int op_id);
/* CallAlloc of `count` operators at once, one host thread each (the model's CONFIG_CONCURRENT_SPAN mode,
 * csrc/core/model/model.cpp:1253-1262); first non-success AsStatus or 0 */
int dihost_ops_alloc_concurrent(dihost_model_t m, const int* op_ids, int count);
/* VirtualCache::GetSeqLength of a request's K cache for one layer (-1: unknown request) */
long dihost_cache_seq_len(dihost_model_t m, int request, int layer);

/* ---- the model runner (dash-infer_amd/host/model_runner.h): AsModel's decode loop over an operator LIST ------------------------
 * dihost_graph_add_op records one OperatorProto of the reference graph (same argument format as dihost_op_create);
 * dihost_graph_build runs the fusion pass (fuse != 0; host/fusion_pass.h) and creates + initialises every operator through the
 * OpFactory.  dihost_graph_report / dihost_graph_fuse_dry (the pass alone, works without a GPU):
 *   "fused=<0|1>;layers=<n>;ops=<before>-><after>;why=<text>;types=<operator types, comma-separated>[;wiring=<op(in)->(out)[weights]|...>]" */
int dihost_graph_add_op(dihost_model_t m, const char* op_type, const char* op_name, const char* inputs, const char* outputs,
                        const char* weights, const char* attrs);
/* The same from a SERIALIZED allspark TransformerProto (csrc/proto/allspark.proto: what the reference's converter writes and AsModel
 * parses, model.cpp:265-287): the operators of the named graphs are appended in order (graphs: comma-separated, NULL = "decoder,gen_graph",
 * the two graphs of a step).  Tensor names and raw attribute bytes are taken as they are (host/graph_wire.h); weights bind by name. */
int dihost_graph_add_serialized(dihost_model_t m, const void* data, size_t bytes, const char* graphs);
int dihost_graph_build(dihost_model_t m, int fuse);
const char* dihost_graph_report(dihost_model_t m);
const char* dihost_graph_fuse_dry(dihost_model_t m);
/* A request enters through its context phase (prompt ids on the host; its cache claims the spans k_spans / v_spans
 * [n_layers][spans_per_req] in order, the first prefix_len tokens already present) and joins the running batch; *first_id = the
 * token sampled after the prompt.  top_k = 1: greedy.  dihost_request_adopt: a request whose cache already holds cached_len tokens
 * joins with next_id as its next input (benchmarks).  dihost_request_stop removes running request `index` (the others move up). */
int dihost_request_start(dihost_model_t m, const int64_t* prompt_host, int len, int prefix_len, int top_k, float top_p, float temperature,
                         unsigned long long seed, int n_layers, int spans_per_req, void* const* k_spans, void* const* v_spans,
                         int64_t* first_id);
int dihost_request_adopt(dihost_model_t m, int cached_len, int64_t next_id, int n_layers, int spans_per_req, void* const* k_spans,
                         void* const* v_spans);
int dihost_request_stop(dihost_model_t m, int index);
/* n decoder steps of the running batch (Alloc -> Forward per operator per step, csrc/core/model/model.cpp:1248-1325);
 * use_graph != 0: the step is captured once as a hipGraph and replayed (fused list only; the context stream must not be NULL).
 * The step's launch plans (attention split count / width, the fused attention block's grid) are made for the running requests' length
 * rounded up to DIHIP_PLAN_BUCKET tokens (default 512; 0: for max_len), never for more than max_len: when the bucket moves the runner
 * synchronises, reshapes the operators and captures the step again (host/model_runner.cpp; HIPContext::PlanLength). */
int dihost_decode_steps(dihost_model_t m, int n, int use_graph);
/* synchronises the stream; ids generated by the last step, one per running request -> count (negative AsStatus on error) */
int dihost_sync_ids(dihost_model_t m, int64_t* ids_host, int capacity);
/* the Request slice behind the id-processing operators (PreProcessId / UpdateId / PostProcessId, host/id_ops_hip.cpp):
 * attach: request `index` of the runtime context gets inputs["input_ids"] = ids [1, len] and the stop conditions UpdateId checks
 * (stop_words: n_words sequences of word_len ids); put_token: what GenerateOp's fill_generated_ids writes after sampling;
 * poll: drains the request's queue of generated tokens -> count, *finish = stop verdict, *n_interim = interim tensors */
int dihost_request_attach(dihost_model_t m, int index, const int64_t* ids, int len, int max_length, int early_stopping, int eos_token_id,
                          const int64_t* stop_words, int n_words, int word_len, int in_length_bias);
int dihost_request_put_token(dihost_model_t m, int index, int position, int64_t token);
int dihost_request_set_step(dihost_model_t m, int index, int step, int in_length_bias);
int dihost_set_phase(dihost_model_t m, int is_context);
int dihost_request_poll(dihost_model_t m, int index, int64_t* tokens, int capacity, int* finish, int* n_interim);
/* GenerateOp's logits processors and log-probability outputs (generate_op.cpp:239-312 build_batch_gencfg, :536-538, :600-650; the
 * GenerateConfig fields of csrc/interface/allspark.h:122-146: neutral values 1 / 0 / 0 / 0 / 0 / - / 0 / 0 / 0).
 *   next_request_generation: for the NEXT request started through dihost_request_start -- the model runner gives it a device-resident token
 *     history (the prompt; every step appends its input id) and, with logprobs, a device-resident record log: the step still replays as a graph;
 *   request_generation:      for request `index` of the runtime context (operator-by-operator use: GenerateOp stages the request's host
 *     "generated_ids" every Forward, as the reference's fill_max_dec_ids copies generated_ids_gpu); input_len = its prompt length;
 *   request_logprobs:        `count` records from `first` -> token_logprob [count], top_value / top_index [count, top_n] (top_n <= 10).  runner != 0:
 *     the running batch's request `index`, records indexed by the token's POSITION in its sequence (first generated token of an L-token prompt:
 *     record L); runner == 0: the runtime context's request, records in generation order.  -> records copied, or a negative AsStatus. */
int dihost_next_request_generation(dihost_model_t m, float repetition_penalty, float frequency_penalty, float presence_penalty,
                                   int no_repeat_ngram_size, int min_length, int eos_token_id, int suppress_repetition_in_generation,
                                   int logprobs, int top_logprobs);
int dihost_request_generation(dihost_model_t m, int index, float repetition_penalty, float frequency_penalty, float presence_penalty,
                              int no_repeat_ngram_size, int min_length, int eos_token_id, int suppress_repetition_in_generation, int logprobs,
                              int top_logprobs, int input_len);
int dihost_request_logprobs(dihost_model_t m, int index, int runner, int first, int count, int top_n, float* token_logprob,
                            float* top_value, int* top_index);
int dihost_running_batch(dihost_model_t m);
/* benchmarks: every running request (and its cache) back to cached_len tokens; spans, shapes and the captured step are kept */
int dihost_requests_rewind(dihost_model_t m, int cached_len);

const char* dihost_last_error(void);
/* "GemmA16W8,GemmA16W4,DecOptMHA,DecOptMQA,AllReduce": op types registered for DeviceType::HIP */
const char* dihost_registered_ops(void);
#ifdef __cplusplus
}
#endif
#endif
