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


# ---- sparse records (CSC / ELL): densified on the way in ---------------------------------------------------------------------------------
SPARSE_GOLDEN = os.path.join(ROOT, "tests", "golden", "tiny_sparse.asparam")


def _dense_of(hostapi, path, name, rank=0, nranks=1, dtype=np.float32):
    raw, shape = hostapi.weight_file_slice(path, name, rank, nranks)
    return np.frombuffer(raw, dtype=dtype).reshape(shape)


def test_sparse_records_written_by_the_reference_writer_are_densified(hostapi):
    """tests/golden/tiny_sparse.asparam: two matrices through the reference writer's CSC and ELL encodings (allsparkz_util.cpp:162-254 over
    sparse_util.cpp:23-131), a VSPLIT CSC matrix, a dense record behind them.  The reader hands out the DENSE matrices the writer started from --
    bit for bit (every non-zero is above the writers' 1e-9 threshold; the zero-valued padding entries the writers add change nothing) -- the index
    steps over the compressed data to the next record, and the tensor-parallel split applies to the densified matrix."""
    mod = _golden_records()
    want = mod.sparse_model()
    got = hostapi.weight_file_index(SPARSE_GOLDEN)
    assert [r[0] for r in got] == [r[0] for r in want]
    for (name, dt, shape, split, off, nb), (wname, arr, wsplit, _, *enc) in zip(got, want):
        assert shape == list(arr.shape) and split == wsplit and nb == arr.nbytes, name   # the DENSE size, whatever the encoding
        dense = _dense_of(hostapi, SPARSE_GOLDEN, name)
        assert dense.tobytes() == arr.tobytes(), name
    name, arr = want[2][0], want[2][1]
    for nranks in (2, 4):
        for rank in range(nranks):
            w = arr.shape[1] // nranks
            share = _dense_of(hostapi, SPARSE_GOLDEN, name, rank, nranks)
            assert share.tobytes() == np.ascontiguousarray(arr[:, rank * w:(rank + 1) * w]).tobytes(), (nranks, rank)


def test_fresh_sparse_files_from_the_reference_writer(hostapi, tmp_path):
    mod = _ref_writer_or_skip()
    rng = np.random.default_rng(8)
    recs = []
    for i, (rows, cols, keep, enc) in enumerate([(64, 40, 0.1, "csc"), (64, 40, 0.1, "ell"), (7, 5, 0.5, "csc"), (12, 9, 0.3, "ell"), (16, 8, 0.0, "csc"),
                                                 (33, 17, 1.0, "ell")]):
        a = (rng.normal(0, 1, (rows, cols)).astype(np.float32) + 4.0) * (rng.random((rows, cols)) < keep)
        recs.append((f"m{i}.{enc}", a.astype(np.float32), 0, False, enc))
    path = str(tmp_path / "sparse.asparam")
    mod.write(path, recs)
    for name, a, *_ in recs:
        assert _dense_of(hostapi, path, name).tobytes() == a.tobytes(), name


def _record(name, descr, shape, sparse_type, nnz, payload):
    head = f"{{'descr': '{descr}', 'fortran_order': False, 'shape': {tuple(shape)},'group_list': (),'sparse_type': {sparse_type},'nnz': {nnz},'split_type': 0,}}\n"
    n = name.encode()
    return b"AS" + (1).to_bytes(2, "little") + len(n).to_bytes(2, "little") + n + head.encode() + payload


def test_f16_sparse_records_and_malformed_sparse_records(hostapi, tmp_path):
    """The f16 forms of both encodings (the reference writes them under ENABLE_FP16: VECT = 8), built here by the layouts of
    allsparkz_util.cpp:184-203 / :234-252; a padding entry with a row index outside the matrix is accepted when its value is zero (the ELL
    writer pads with whatever index its buffer held, sparse_util.cpp:111), refused when it is not."""
    rows, cols, vect = 20, 3, 8
    dense = np.zeros((rows, cols), np.float16)
    dense[[1, 4, 19], 0] = [1.5, -2.0, 0.25]
    dense[7, 2] = 3.0
    # CSC: per column the non-zeros, padded to a multiple of VECT with (last row, 0)
    off, ridx, val = [0], [], []
    for c in range(cols):
        r = [i for i in range(rows) if dense[i, c] != 0]
        v = [dense[i, c] for i in r]
        while len(r) % vect:
            r.append(r[-1])
            v.append(np.float16(0))
        ridx += r
        val += v
        off.append(len(ridx))
    csc = np.asarray(off, np.int32).tobytes() + np.asarray(ridx, np.int32).tobytes() + np.asarray(val, np.float16).tobytes()
    # ELL: max_c = 8 entries per column, one block of VECT: for col: 8 entries; padding rows 65535 (outside) with value 0
    eidx, eval_ = [], []
    for c in range(cols):
        r = [i for i in range(rows) if dense[i, c] != 0]
        v = [dense[i, c] for i in r]
        eidx += r + [65535] * (vect - len(r))
        eval_ += v + [np.float16(0)] * (vect - len(v))
    ell = np.asarray(eidx, np.uint16).tobytes() + np.asarray(eval_, np.float16).tobytes()
    end = b"AS\x00\x00\x00\x00"
    path = str(tmp_path / "f16.asparam")
    open(path, "wb").write(_record("csc16", "<f2", (rows, cols), 1, len(ridx), csc) + _record("ell16", "<f2", (rows, cols), 2, len(eidx), ell) + end)
    for name in ("csc16", "ell16"):
        assert _dense_of(hostapi, path, name, dtype=np.float16).tobytes() == dense.tobytes(), name

    def refused(data, needle):
        p = str(tmp_path / "bad_sparse.asparam")
        open(p, "wb").write(data)
        with pytest.raises(hostapi.HostError) as e:
            hostapi.weight_file_index(p)
            hostapi.weight_file_slice(p, "x", 0, 1)
        assert needle in str(e.value), str(e.value)

    bad_idx = np.asarray([0, 8, 8, 8], np.int32).tobytes() + np.asarray([99] * 8, np.int32).tobytes() + np.asarray([1.0] * 8, np.float16).tobytes()
    refused(_record("x", "<f2", (rows, cols), 1, 8, bad_idx) + end, "outside the matrix")            # a NON-zero entry outside the matrix
    refused(_record("x", "<f2", (rows, cols), 1, 8, csc) + end, "")                                   # nnz disagrees with the stored data
    refused(_record("x", "<f2", (rows, cols), 2, 10, ell) + end, "ELL")                               # nnz not cols x a multiple of VECT
    refused(_record("x", "<i4", (rows, cols), 1, 8, csc) + end, "sparse")                             # not a floating-point matrix
    refused(_record("x", "<f2", (rows, cols), 3, 8, csc) + end, "sparse_type")
