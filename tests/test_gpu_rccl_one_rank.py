"""The RCCL wrappers of include/dashinfer_hip.h section 6 EXECUTED ("zero executions in any test"): a one-GPU box cannot
hold two RCCL ranks (one device per rank), so the communicator here has ONE rank -- ncclGetUniqueId -> ncclCommInitRank(1, id, 0) ->
ncclAllReduce / ncclAllGather on the caller's stream: sum over one rank == the input, for f32 / f16 / bf16, out of place and in place, also
captured into a hipGraph and replayed (the decode step's form), and through the AllReduce operator of the C++ layer with the context's
GetRCCLComm() set (allreduce_op.cpp:23-95; DIHIP_ALLREDUCE_FORCE_RCCL=1 keeps the one-rank op off its copy shortcut)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}   # DIHIP_F32 / DIHIP_F16 / DIHIP_BF16 (span::DataType numbering)


@pytest.fixture(scope="module")
def comm(pkg):
    from dash_infer_amd import capi
    lib = capi.lib()
    uid = C.create_string_buffer(128)
    assert lib.dihip_rccl_unique_id(uid) == 0, lib.dihip_last_error()
    assert any(b != 0 for b in uid.raw)
    h = C.c_void_p()
    assert lib.dihip_rccl_comm_init_rank(C.byref(h), 1, uid, 0) == 0, lib.dihip_last_error()
    assert h.value
    yield lib, h
    assert lib.dihip_rccl_comm_destroy(h) == 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("count", [3584, 32 * 3584, 152064, 7])
def test_allreduce_sum_over_one_rank_is_the_input(comm, dtype, count):
    from dash_infer_amd import ops
    lib, h = comm
    g = torch.Generator(device="cuda").manual_seed(count)
    x = torch.randn(count, device="cuda", generator=g).to(dtype)
    y = torch.full_like(x, 7.0)
    st = ops.cur_stream()
    assert lib.dihip_allreduce_sum(h, st, x.data_ptr(), y.data_ptr(), count, DT[dtype]) == 0, lib.dihip_last_error()
    torch.cuda.synchronize()
    assert torch.equal(x, y)
    keep = x.clone()
    assert lib.dihip_allreduce_sum(h, st, x.data_ptr(), x.data_ptr(), count, DT[dtype]) == 0   # in place, the decode step's form
    torch.cuda.synchronize()
    assert torch.equal(x, keep)


def test_allreduce_refuses_what_the_reference_refuses(comm):
    lib, h = comm
    x = torch.zeros(16, dtype=torch.int32, device="cuda")
    assert lib.dihip_allreduce_sum(h, None, x.data_ptr(), x.data_ptr(), 16, 5) != 0      # not an FT code: GetNcclType has no entry (nccl_utils.hpp:9-27)
    assert lib.dihip_allreduce_sum(None, None, x.data_ptr(), x.data_ptr(), 16, 0) != 0   # no communicator
    assert lib.dihip_allreduce_sum(h, None, x.data_ptr(), x.data_ptr(), 0, 0) == 0       # empty message: nothing to do


def test_allreduce_inside_a_captured_graph(comm):
    """The decode step replays its all-reduces from a hipGraph: capture two collectives around an elementwise kernel, replay thrice."""
    from dash_infer_amd import ops
    lib, h = comm
    x = torch.arange(3584, device="cuda", dtype=torch.float32).to(torch.bfloat16)
    y = torch.zeros_like(x)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        assert lib.dihip_allreduce_sum(h, ops.cur_stream(), x.data_ptr(), y.data_ptr(), x.numel(), 2) == 0   # warm: RCCL's lazy set-up outside capture
    s.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            assert lib.dihip_allreduce_sum(h, ops.cur_stream(), x.data_ptr(), y.data_ptr(), x.numel(), 2) == 0, lib.dihip_last_error()
            y.mul_(2)
            assert lib.dihip_allreduce_sum(h, ops.cur_stream(), y.data_ptr(), y.data_ptr(), x.numel(), 2) == 0
    for i in range(3):
        x.fill_(float(i + 1))
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(y, torch.full_like(y, 2.0 * (i + 1)))


def test_allgather_over_one_rank(comm):
    from dash_infer_amd import ops
    lib, h = comm
    x = torch.randn(5, 96, device="cuda").to(torch.bfloat16)
    out = torch.zeros_like(x)
    assert lib.dihip_allgather_bytes(h, ops.cur_stream(), x.data_ptr(), out.data_ptr(), x.numel() * 2) == 0, lib.dihip_last_error()
    torch.cuda.synchronize()
    assert torch.equal(x, out)
    out.zero_()
    tmp = torch.zeros_like(x)
    assert lib.dihip_allgather_rows(h, ops.cur_stream(), x.data_ptr(), tmp.data_ptr(), out.data_ptr(), 5, 96 * 2, 1) == 0
    torch.cuda.synchronize()
    assert torch.equal(x, out)


@pytest.mark.parametrize("dtype,code", [(torch.bfloat16, "bf16"), (torch.float16, "f16"), (torch.float32, "f32")])
def test_allreduce_operator_with_the_contexts_rccl_communicator(comm, monkeypatch, dtype, code):
    """AllReduceOpHIP (REGISTER_OP(AllReduce, HIP)) with HIPContext::GetRCCLComm() set, as the engine's rank thread sets it
    (cuda_context.cpp:154-169): Init -> Reshape -> Forward goes through dihip_allreduce_sum, out of place and in place."""
    from dash_infer_amd import hostapi, ops
    lib, h = comm
    monkeypatch.setenv("DIHIP_ALLREDUCE_FORCE_RCCL", "1")
    m = hostapi.Model(ops.cur_stream(), 4, 2, 128, 16, rank=0, nranks=1, comm=h)
    try:
        x = torch.randn(3, 1, 3584, device="cuda").to(dtype)
        m.set_tensor("x", x, code)
        op = m.create_op("AllReduce", "decoder.layer.0.attention.all_reduce", ["x"], ["y"])
        m.reshape(op)
        m.forward(op)
        torch.cuda.synchronize()
        dt, shape, ptr = m.get_tensor("y")
        assert shape == [3, 1, 3584]
        from tests.test_gpu_host_ops import view_of
        assert torch.equal(view_of(ptr, shape, dtype), x)
        op2 = m.create_op("AllReduce", "decoder.layer.0.ffn.all_reduce", ["x"], ["x"])    # in place, as the graph wires it
        m.reshape(op2)
        m.forward(op2)
        torch.cuda.synchronize()
        assert torch.isfinite(x.float()).all()
    finally:
        m.close()
