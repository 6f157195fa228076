// allreduce_op_hip.cpp -- op type "AllReduce" on DeviceType::HIP (AllReduceOp,
// csrc/core/operator/nccl/allreduce/allreduce_op.cpp:23-95): in -> out sum all-reduce over the RCCL
// communicator of the context.  Unlike the reference (which calls ctx->Synchronize() after the
// collective, allreduce_op.cpp:90) the op only enqueues: ordering is the stream's.
#include <algorithm>
#include <cstdlib>

#include "dashinfer_hip.h"
#include "operator.h"

namespace allspark {

class AllReduceOpHIP : public AsOperator {
 public:
  explicit AllReduceOpHIP(const std::string& op_type = "") : AsOperator(op_type) {}
  AsStatus Init(const OperatorProto& op_proto, const DeviceContext& ctx, const TensorMap& weights_map,
                TensorMap* tensor_map) override {
    AS_CHECK_STATUS(AsOperator::Init(op_proto, ctx, weights_map, tensor_map));
    tensor_map_->at(out_names_[0])->SetDataType(tensor_map_->at(in_names_[0])->GetDataType());
    return AsStatus::ALLSPARK_SUCCESS;
  }
  AsStatus Reshape(RuntimeContext*) override {
    AsTensor* x = tensor_map_->at(in_names_[0]).get();
    count_ = x->Count();
    return tensor_map_->at(out_names_[0])->SetShape(Shape(x->GetShape()));
  }
  ~AllReduceOpHIP() override {
    if (fork_) (void)hipEventDestroy(fork_);
    if (join_) (void)hipEventDestroy(join_);
  }
  AsStatus Forward(RuntimeContext* rt) override {
    const HIPContext* h = static_cast<const HIPContext*>(ctx_);
    // DIHIP_TP_OVERLAP=1 (north star): the collective on the context's side stream between two events, the weights of the operator
    // that reads the reduced rows pulled on-die by the main stream meanwhile (HIPContext::ConsumerWeights), then the join.  Under
    // stream capture the events become a fork / join of the graph.  Decoder phase only (the context phase's messages are bandwidth).
    static const bool overlap = [] { const char* e = getenv("DIHIP_TP_OVERLAP"); return e && e[0] == '1'; }();
    if (overlap && h->GetNranks() > 1 && rt && !rt->is_context && !in_side_) {
      hipStream_t main = h->GetStream(), side = h->SideStream();
      if (side && (fork_ || hipEventCreateWithFlags(&fork_, hipEventDisableTiming) == hipSuccess) &&
          (join_ || hipEventCreateWithFlags(&join_, hipEventDisableTiming) == hipSuccess)) {
        if (hipEventRecord(fork_, main) != hipSuccess || hipStreamWaitEvent(side, fork_, 0) != hipSuccess) return AsStatus::ALLSPARK_RUNTIME_ERROR;
        const_cast<HIPContext*>(h)->SetStream(side);
        in_side_ = true;
        const AsStatus st = Forward(rt);  // the collective itself, on the side stream
        in_side_ = false;
        const_cast<HIPContext*>(h)->SetStream(main);
        AS_CHECK_STATUS(st);
        if (hipEventRecord(join_, side) != hipSuccess) return AsStatus::ALLSPARK_RUNTIME_ERROR;
        if (const auto* ws = h->ConsumerWeights(out_names_[0])) {
          const void* bufs[8];
          size_t bytes[8];
          int n = 0;
          for (const auto& w : *ws)
            if (n < 8) {
              bufs[n] = w.ptr;
              bytes[n++] = w.bytes;
            }
          if (n) AS_CHECK_STATUS(FromDihip(dihip_prefetch(main, bufs, bytes, n, 64)));
        }
        return hipStreamWaitEvent(main, join_, 0) == hipSuccess ? AsStatus::ALLSPARK_SUCCESS : AsStatus::ALLSPARK_RUNTIME_ERROR;
      }
    }
    AsTensor* x = tensor_map_->at(in_names_[0]).get();
    AsTensor* y = tensor_map_->at(out_names_[0]).get();
    // single rank: copy (allreduce_op.cpp:70-80 behaviour) -- unless DIHIP_ALLREDUCE_FORCE_RCCL=1 sends a one-rank communicator through the
    // collective itself (the only way a one-GPU box executes the RCCL path: tests/test_gpu_rccl_one_rank.py)
    const char* fe = h->GetNranks() == 1 ? getenv("DIHIP_ALLREDUCE_FORCE_RCCL") : nullptr;
    const bool force_rccl = fe && fe[0] == '1';
    if (h->GetNranks() == 1 && !(force_rccl && h->GetRCCLComm())) {
      if (x->GetDataPtr() != y->GetDataPtr() &&
          hipMemcpyAsync(y->GetDataPtr(), x->GetDataPtr(), x->GetSizeInByte(), hipMemcpyDeviceToDevice, h->GetStream()) != hipSuccess)
        return AsStatus::ALLSPARK_RUNTIME_ERROR;
      return AsStatus::ALLSPARK_SUCCESS;
    }
    const size_t bytes = (size_t)count_ * SizeofType(x->GetDataType());
    if (h->GetP2PComm() && bytes <= dihip_p2p_ar_max_bytes() && bytes % 16 == 0)  // decode rows: latency bound on a ring, one shot over xGMI instead
      return FromDihip(dihip_p2p_allreduce_sum(h->GetP2PComm(), h->GetStream(), x->GetDataPtr(), y->GetDataPtr(), (size_t)count_,
                                               DihipDtype(x->GetDataType())));
    if (h->GetP2PComm() && !h->GetRCCLComm() && bytes % 16 == 0) {
      // no RCCL communicator (rank threads on one GPU; a node set up with the peer-to-peer path only): a longer message -- the K-split
      // lm_head's logits rows -- goes out as slot-sized pieces, each a one-shot exchange of its own
      const size_t es = SizeofType(x->GetDataType()), piece = dihip_p2p_ar_max_bytes() / es;
      for (size_t at = 0; at < (size_t)count_; at += piece) {
        const size_t n = std::min(piece, (size_t)count_ - at);
        AS_CHECK_STATUS(FromDihip(dihip_p2p_allreduce_sum(h->GetP2PComm(), h->GetStream(), (const char*)x->GetDataPtr() + at * es,
                                                          (char*)y->GetDataPtr() + at * es, n, DihipDtype(x->GetDataType()))));
      }
      return AsStatus::ALLSPARK_SUCCESS;
    }
    if (!h->GetRCCLComm()) return AsStatus::ALLSPARK_PARAM_ERROR;
    return FromDihip(dihip_allreduce_sum(h->GetRCCLComm(), h->GetStream(), x->GetDataPtr(), y->GetDataPtr(), (size_t)count_,
                                         DihipDtype(x->GetDataType())));
  }

 private:
  int64_t count_ = 0;
  hipEvent_t fork_ = nullptr, join_ = nullptr;
  bool in_side_ = false;
};
REGISTER_OP(AllReduce, HIP, AllReduceOpHIP)

}  // namespace allspark
