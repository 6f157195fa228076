"""The reader of the reference's serialized weights (dash-infer_amd/host/weight_file.h, `dihost_weight_file_index`; ".asparam", the container the
converter writes: csrc/utility/allsparkz_util.cpp:264-339, read on the reference side by WeightFileParser, weight_loader.cpp:20-130) against files
written by the REFERENCE'S OWN writer:

  * tests/golden/tiny_qwen2_a16w4.asparam (committed; tests/golden/make_asparam_golden.py): every record's name, element type, shape, split mode,
    and the bytes at the offset the index gives;
  * where oracle/_ref/libdashinfer_ref_asparam.so exists (built from /root/reference by oracle/Makefile), fresh files: every element type of the
    loader's table, rank-1 / rank-3 shapes, long names, an empty file (global header only);
  * malformed containers are refused as a whole: truncated data, a missing global header, a bad magic, a sparse encoding.
No GPU: indexing is host work (the upload is tests/test_gpu_host_ops.py::test_weights_from_a_serialized_file)."""
import os
import shutil

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "tiny_qwen2_a16w4.asparam")
# allspark.proto DataType
DT = {"float32": 1, "float16": 2, "int8": 3, "int16": 4, "int32": 5, "int64": 6, "bool": 8, "bf16": 9, "uint8": 10}


@pytest.fixture(scope="module")
def hostapi(pkg):
    from dash_infer_amd import hostapi as h
    return h


def _golden_records():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_asparam_golden", os.path.join(ROOT, "tests", "golden", "make_asparam_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_golden_file_written_by_the_reference_writer(hostapi):
    mod = _golden_records()
    want = mod.tiny_model()
    got = hostapi.weight_file_index(GOLDEN)
    assert [r[0] for r in got] == [r[0] for r in want], "record names / order"
    raw = open(GOLDEN, "rb").read()
    for (name, dt, shape, split, off, nb), (wname, arr, wsplit, bf16) in zip(got, want):
        assert dt == (DT["bf16"] if bf16 else DT[str(arr.dtype)]), name
        assert shape == list(arr.shape) and split == wsplit and nb == arr.nbytes, name
        assert raw[off:off + nb] == np.ascontiguousarray(arr).tobytes(), f"{name}: bytes at offset {off}"
    # the container ends right behind the last record with the global header "AS" 0 0
    last = got[-1]
    assert raw[last[4] + last[5]:] == b"AS\x00\x00\x00\x00"


def _ref_writer_or_skip():
    path = os.path.join(ROOT, "oracle", "_ref", "libdashinfer_ref_asparam.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libdashinfer_ref_asparam.so not built (needs /root/reference: oracle/Makefile refasparam)")
    return _golden_records()


def test_fresh_files_every_element_type(hostapi, tmp_path):
    mod = _ref_writer_or_skip()
    rng = np.random.default_rng(1)
    recs = [("f32.matrix", rng.normal(size=(3, 5)).astype(np.float32), 0, False),
            ("f16.vector", rng.normal(size=(7,)).astype(np.float16), 1, False),
            ("bf16.as.bits", rng.integers(0, 65536, (4, 2, 3), dtype=np.uint16), 2, True),
            ("i8", rng.integers(-128, 128, (16, 16), dtype=np.int8), 5, False),
            ("u8.nibbles", rng.integers(0, 256, (128, 8), dtype=np.uint8), 6, False),
            ("i32", np.arange(6, dtype=np.int32).reshape(2, 3), 0, False),
            ("i64.one", np.array([2 ** 40], np.int64), 0, False),
            ("x" * 300 + ".long.name", np.zeros((2, 2), np.float32), 11, False)]
    path = str(tmp_path / "fresh.asparam")
    mod.write(path, recs)
    got = hostapi.weight_file_index(path)
    raw = open(path, "rb").read()
    assert len(got) == len(recs)
    for (name, dt, shape, split, off, nb), (wname, arr, wsplit, bf16) in zip(got, recs):
        assert name == wname and dt == (DT["bf16"] if bf16 else DT[str(arr.dtype)]) and shape == list(arr.shape) and split == wsplit
        assert raw[off:off + nb] == arr.tobytes()
    empty = str(tmp_path / "empty.asparam")
    mod.write(empty, [])
    assert hostapi.weight_file_index(empty) == []


def test_malformed_containers_are_refused(hostapi, tmp_path):
    raw = open(GOLDEN, "rb").read()

    def refused(data, needle):
        p = str(tmp_path / "bad.asparam")
        open(p, "wb").write(data)
        with pytest.raises(hostapi.HostError) as e:
            hostapi.weight_file_index(p)
        assert needle in str(e.value), str(e.value)

    refused(raw[:-6], "global header")                       # the end marker is missing
    refused(raw[:len(raw) // 2], "")                         # cut in the middle of a data block: bad magic or truncation, never a partial index
    refused(b"XS" + raw[2:], "magic")
    refused(raw.replace(b"'sparse_type': 0", b"'sparse_type': 1", 1), "sparse")
    refused(raw.replace(b"'descr': '<b2'", b"'descr': '>b2'", 1), "big-endian")
    refused(raw.replace(b"'descr': '<b2'", b"'descr': '<c8'", 1), "element type")
    with pytest.raises(hostapi.HostError):
        hostapi.weight_file_index(str(tmp_path / "does.not.exist"))
