"""The tensor-parallel split of a serialized weight file AT LOAD TIME (VERDICT r5 missing #1): dihost_weights_load_file applies every
record's SplitMode + group_list for the model's (rank, nranks) -- host/weight_file.h SliceForRank, the reference's WeightManager -> WeightSplitter
(csrc/runtime/weight/weight_splitter.cpp:60-127 VSPLIT, :369-438 HSPLIT, :611-721 GROUP_VSPLIT, :128-232 / :439-520 BATCH_V/HSPLIT, :521-610
QKVSPLIT, :722-852 MQA_VSPLIT, :853-919 EPSPLIT).  On the host, no GPU (`dihost_weight_file_slice` is what the loader uploads):

  * tests/golden/tiny_qwen2_a16w4_tp.asparam -- written by the REFERENCE'S OWN writer with the group_lists its converter gives
    (tests/golden/make_asparam_golden.py tp_model) -- every record x TP 1 / 2 / 4 x every rank against oracle/weight_split.py;
  * the Qwen2 layer tensors against dash-infer_amd/tp.py's slices (round 2's independent statement of the same rules): qkv columns by the rank's
    heads, o rows, gate / up columns, down rows, sub-channel parameters along the groups;
  * what does not divide, a GROUP_VSPLIT without its group_list and a SplitMode without a splitter are REFUSED with the record's name;
  * a header whose shape would overflow the byte count is refused when the file is indexed."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import weight_split as ws

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "tiny_qwen2_a16w4_tp.asparam")


@pytest.fixture(scope="module")
def hostapi(pkg):
    from dash_infer_amd import hostapi as h
    return h


def _maker():
    spec = importlib.util.spec_from_file_location("make_asparam_golden", os.path.join(ROOT, "tests", "golden", "make_asparam_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("nranks", [1, 2, 4])
def test_every_record_of_the_golden_export_against_the_oracle(hostapi, nranks):
    recs = _maker().tp_model()
    index = {r[0]: r for r in hostapi.weight_file_index(GOLDEN)}
    assert list(index) == [r[0] for r in recs]
    for name, arr, mode, bf16, groups in recs:
        assert index[name][3] == mode
        for rank in range(nranks):
            if name == "experts.ep.weight" and 8 % nranks:
                continue
            want = ws.split(arr, mode, rank, nranks, groups)
            got, shape = hostapi.weight_file_slice(GOLDEN, name, rank, nranks)
            assert shape == list(want.shape), (name, rank, shape, want.shape)
            assert got == np.ascontiguousarray(want).tobytes(), f"{name}: rank {rank} of {nranks}"
    # the shares of a column- / row-split tensor tile the whole
    for name, axis in (("decoder.layer.0.ffn.intermediate.dense.weight", 1), ("decoder.layer.0.ffn.output.dense.weight", 0), ("lm_head.weight", 1)):
        arr = next(r[1] for r in recs if r[0] == name)
        parts = [np.frombuffer(hostapi.weight_file_slice(GOLDEN, name, r, nranks)[0], arr.dtype).reshape(hostapi.weight_file_slice(GOLDEN, name, r, nranks)[1])
                 for r in range(nranks)]
        assert np.array_equal(np.concatenate(parts, axis), arr)
    # a row-parallel bias lives on rank 0 alone
    b = next(r[1] for r in recs if r[0].endswith("output.dense.bias"))
    for rank in range(nranks):
        got = np.frombuffer(hostapi.weight_file_slice(GOLDEN, "decoder.layer.0.attention.output.dense.bias", rank, nranks)[0], np.uint16)
        assert np.array_equal(got, b if rank == 0 else np.zeros_like(b))


@pytest.mark.parametrize("nranks", [2])
def test_layer_tensors_equal_tp_py_slices(hostapi, pkg, nranks):
    """dash-infer_amd/tp.py (what DecodeSession / bench.py --gpus N shard with): the rank's query / KV heads -> qkv columns, o rows; FFN
    columns in units of the quantisation group -> gate / up columns, down rows."""
    from dash_infer_amd import tp
    recs = {r[0]: r for r in _maker().tp_model()}
    n, g, H, inter, G = 4, 2, 128, 512, 128
    shards = tp.shard_heads(n, g, nranks)
    ffn = tp.shard_ffn(inter, nranks, G)
    p = "decoder.layer.0."
    for rank in range(nranks):
        cols = tp.qkv_columns(shards[rank], n, g, H)
        sc = recs[p + "attention.self.weight.scale"][1]
        got = np.frombuffer(hostapi.weight_file_slice(GOLDEN, p + "attention.self.weight.scale", rank, nranks)[0], np.uint16).reshape(sc.shape[0], -1)
        assert np.array_equal(got, sc[:, cols])
        w4 = recs[p + "attention.self.weight"][1]       # nibble pairs: byte j = columns 2j, 2j + 1
        got = np.frombuffer(hostapi.weight_file_slice(GOLDEN, p + "attention.self.weight", rank, nranks)[0], np.uint8).reshape(w4.shape[0], -1)
        assert np.array_equal(got, w4[:, [c // 2 for c in cols[::2]]])
        rows = tp.o_rows(shards[rank], H)
        ow = recs[p + "attention.output.dense.weight"][1]
        got = np.frombuffer(hostapi.weight_file_slice(GOLDEN, p + "attention.output.dense.weight", rank, nranks)[0], np.uint8).reshape(-1, ow.shape[1])
        assert np.array_equal(got, ow[rows])
        osc = recs[p + "attention.output.dense.weight.scale"][1]   # sub-channel parameters of a row-split weight: split along the groups
        got = np.frombuffer(hostapi.weight_file_slice(GOLDEN, p + "attention.output.dense.weight.scale", rank, nranks)[0], np.uint16).reshape(-1, osc.shape[1])
        assert np.array_equal(got, osc[[r_ // G for r_ in rows[::G]]])
        fr = list(ffn[rank])
        gw = recs[p + "ffn.intermediate.dense.weight.scale"][1]
        got = np.frombuffer(hostapi.weight_file_slice(GOLDEN, p + "ffn.intermediate.dense.weight.scale", rank, nranks)[0], np.uint16).reshape(gw.shape[0], -1)
        assert np.array_equal(got, gw[:, fr])
        dw = recs[p + "ffn.output.dense.weight"][1]
        got = np.frombuffer(hostapi.weight_file_slice(GOLDEN, p + "ffn.output.dense.weight", rank, nranks)[0], np.uint8).reshape(-1, dw.shape[1])
        assert np.array_equal(got, dw[fr])


def test_what_the_reference_refuses_is_refused_with_the_records_name(hostapi, tmp_path):
    lib_path = os.path.join(ROOT, "oracle", "_ref", "libdashinfer_ref_asparam.so")
    if not os.path.exists(lib_path):
        pytest.skip("oracle/_ref/libdashinfer_ref_asparam.so not built (needs /root/reference)")
    mod = _maker()
    path = str(tmp_path / "bad.asparam")
    z = lambda *s: np.zeros(s, np.float32)
    mod.write(path, [("v.odd", z(4, 6), 1, False, []), ("h.odd", z(6, 4), 2, False, []), ("g.nolist", z(4, 8), 6, False, []),
                     ("g.badsum", z(4, 8), 6, False, [4, 2]), ("g.odd", z(4, 8), 6, False, [6, 2]), ("kv.batch", z(2, 4, 8), 10, False, []),
                     ("hq", z(4, 8), 5, False, []), ("bh.2d", z(4, 8), 9, False, []), ("ep.2d", z(4, 8), 11, False, []), ("fine", z(4, 8), 1, False, [])])
    for name in ("v.odd", "h.odd", "g.nolist", "g.badsum", "g.odd", "kv.batch", "hq", "bh.2d", "ep.2d"):
        with pytest.raises(hostapi.HostError, match=name.replace(".", r"\.")):
            hostapi.weight_file_slice(path, name, 1, 4)
        hostapi.weight_file_slice(path, name, 0, 1)                # one rank: everything is whole
    assert hostapi.weight_file_slice(path, "fine", 3, 4)[1] == [4, 2]
    with pytest.raises(hostapi.HostError, match="no record named"):
        hostapi.weight_file_slice(path, "absent", 0, 2)
    with pytest.raises(hostapi.HostError):
        hostapi.weight_file_slice(path, "fine", 4, 4)              # rank outside the group
    for name, arr, mode, groups in (("v.odd", z(4, 6), 1, []), ("g.odd", z(4, 8), 6, [6, 2]), ("kv.batch", z(2, 4, 8), 10, [])):
        with pytest.raises(ws.NotSplittable):
            ws.split(arr, mode, 1, 4, groups)                       # the oracle refuses the same


def test_a_shape_that_overflows_the_byte_count_is_refused(hostapi, tmp_path):
    """`count *= d` on a malformed header must not wrap into a small or negative byte count."""
    good = open(GOLDEN, "rb").read()
    key = b"'shape': (320, 256)"
    assert key in good
    for bad_shape in (b"'shape': (4294967296, 4294967296)", b"'shape': (9223372036854775807, 3)", b"'shape': (3200000, 2560000)"):
        path = str(tmp_path / "overflow.asparam")
        open(path, "wb").write(good.replace(key, bad_shape, 1))
        with pytest.raises(hostapi.HostError, match="larger than the file|truncated"):
            hostapi.weight_file_index(path)
