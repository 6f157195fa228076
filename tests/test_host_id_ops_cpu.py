"""The graph-head / graph-tail operators of the reference's generation graph on DeviceType::HIP (dash-infer_amd/host/id_ops_hip.cpp),
host logic on CPU: TransMask (no-op under a flash-style prefill, transmask_op.cpp:101-107), PostProcessId (no-op,
postprocess_id_op.cpp:27-31), UpdateId without PreProcessId's tensors (refused) and its "update_id_first" form (returns at once,
update_id_op.cpp:144).  PreProcessId copies to the device: its test and UpdateId's stop conditions are in tests/test_gpu_host_ops.py."""
import pytest


@pytest.fixture()
def model(pkg):
    from dash_infer_amd import hostapi
    m = hostapi.Model(None, 4, 2, 128, 16, max_batch=2, max_len=64)
    yield m
    m.close()


def _rt(m, steps):
    # runtime context with len(steps) requests and no cache spans (the id operators do not touch the cache)
    m.set_runtime(False, steps, [[[]] for _ in steps], [[[]] for _ in steps])


def test_transmask_and_postprocess_are_accepted_and_compute_nothing(model):
    from dash_infer_amd import hostapi
    tm = model.create_op("TransMask", "transmask", ["input_ids"], ["attention_mask"], [], "sequence_mask=b:1")
    model.reshape(tm)
    model.forward(tm)
    with pytest.raises(hostapi.HostError):      # sequence_mask + blank: refused like the reference (transmask_op.cpp:64-68)
        model.create_op("TransMask", "transmask2", ["input_ids"], ["attention_mask2"], [], "sequence_mask=b:1;blank=b:1")
    pp = model.create_op("PostProcessId", "postprocess_id", ["generated_ids"], ["generated_ids_out"])
    _rt(model, [0])
    model.reshape(pp)
    model.forward(pp)


def test_update_id_needs_the_request_tensors_preprocess_id_creates(model):
    from dash_infer_amd import hostapi
    up = model.create_op("UpdateId", "update_id", ["x"], ["y"])
    first = model.create_op("UpdateId", "update_id_first", ["x"], ["y2"])
    _rt(model, [5, 5])
    for i in range(2):
        model.request_attach(i, [11, 12, 13, 14, 15], max_length=9, early_stopping=True, eos=99, stop_words=[[7, 8]])
    with pytest.raises(hostapi.HostError):     # interim["generated_ids"] does not exist before PreProcessId ran
        model.forward(up)
    model.forward(first)                        # "update_id_first" returns at once
    toks, finish, n_interim = model.request_poll(0)
    assert toks == [] and not finish and n_interim == 0
