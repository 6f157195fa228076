#!/usr/bin/env python3
"""tests/golden/tiny_qwen2_a16w4.asparam: a serialized weight file written by the REFERENCE'S OWN writer -- allspark::util::
save_allsparky_tofile + set_global_header of csrc/utility/allsparkz_util.cpp, compiled from where it lies into
oracle/_ref/libdashinfer_ref_asparam.so (oracle/Makefile target refasparam) -- one record per weight of a 1-layer Qwen2-shaped model with
A16W4 sub-channel weights, under the names and split modes the reference's converter gives them (qwen_v15.py:130-165, 540-569;
quantization_utils.py:56-110).  The arrays are a pure function of the seed (tiny_model(): tests rebuild them to compare).
Runs only where /root/reference exists (the writer is the reference's); the committed bytes travel.

    python tests/golden/make_asparam_golden.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.dirname(os.path.abspath(__file__))
# allspark.proto SplitMode
NOSPLIT, VSPLIT, HSPLIT, GROUP_VSPLIT = 0, 1, 2, 6


def ref_writer():
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libdashinfer_ref_asparam.so"))
    lib.ref_asparam_append.restype = C.c_int
    lib.ref_asparam_append.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_int64, C.c_char, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int]
    lib.ref_asparam_append_groups.restype = C.c_int
    lib.ref_asparam_append_groups.argtypes = lib.ref_asparam_append.argtypes + [C.POINTER(C.c_int), C.c_int]
    lib.ref_asparam_append_sparse.restype = C.c_int
    lib.ref_asparam_append_sparse.argtypes = lib.ref_asparam_append.argtypes + [C.c_int]
    lib.ref_asparam_finish.restype = C.c_int
    lib.ref_asparam_finish.argtypes = [C.c_char_p]
    return lib


def descr(a, bf16=False):
    """(type letter, word size) as the converter writes them; bf16 arrays travel as uint16 here"""
    if bf16:
        return b"b", 2
    return {"float32": (b"f", 4), "float16": (b"f", 2), "int8": (b"i", 1), "uint8": (b"u", 1), "int64": (b"i", 8), "int32": (b"i", 4)}[str(a.dtype)]


def write(path, records):
    """records: [(name, array, split_mode, bf16?[, group_list])]; a group_list entry "csc" / "ell" instead: the writer's sparse encoding"""
    lib = ref_writer()
    if os.path.exists(path):
        os.remove(path)
    for rec in records:
        name, a, split, bf16 = rec[:4]
        sparse = {"csc": 1, "ell": 2}.get(rec[4], 0) if len(rec) > 4 and isinstance(rec[4], str) else 0
        groups = list(rec[4]) if len(rec) > 4 and not sparse else []
        a = np.ascontiguousarray(a)
        letter, word = descr(a, bf16)
        shape = (C.c_int * a.ndim)(*a.shape)
        if sparse:
            assert lib.ref_asparam_append_sparse(path.encode(), name.encode(), a.ctypes.data, a.nbytes, letter, word, shape, a.ndim, split, sparse) >= 0
        elif groups:
            gl = (C.c_int * len(groups))(*groups)
            assert lib.ref_asparam_append_groups(path.encode(), name.encode(), a.ctypes.data, a.nbytes, letter, word, shape, a.ndim, split, gl,
                                                 len(groups)) == 0
        else:
            assert lib.ref_asparam_append(path.encode(), name.encode(), a.ctypes.data, a.nbytes, letter, word, shape, a.ndim, split) == 0
    assert lib.ref_asparam_finish(path.encode()) == 0


def tp_model(seed=9):
    """The records of a 1-layer Qwen2-shaped export AS THE CONVERTER WRITES THEM for tensor parallelism (qwen_v15.py:130-165, 540-569;
    model_base.py save_torch_to_allsparky): 4 query / 2 KV heads of 128, hidden 256, intermediate 512, vocabulary 320, A16W4 g128 --
    qkv GROUP_VSPLIT with group_list [n H, g H, g H] (halved for the nibble-packed weight), o / down HSPLIT with sub-channel parameters
    HSPLIT along the groups, gate / up VSPLIT, a per-channel int8 o-projection whose parameters stay NOSPLIT, an o bias (HSPLIT vector:
    rank 0 only), plus one tensor per remaining splitter (QKVSPLIT, MQA_VSPLIT, BATCH_V/HSPLIT, EPSPLIT).  [(name, array, split, bf16, groups)]"""
    NOSPLIT, VSPLIT, HSPLIT, QKVSPLIT, MQA_VSPLIT, GROUP_VSPLIT, BATCH_VSPLIT, BATCH_HSPLIT, EPSPLIT = 0, 1, 2, 3, 7, 6, 8, 9, 11
    rng = np.random.default_rng(seed)
    hidden, n, g, H, inter, vocab, G = 256, 4, 2, 128, 512, 320, 128
    bf = lambda shape, s=0.05: (rng.normal(0, s, shape).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
    recs = [("embedding.word_embeddings", bf((vocab, hidden)), NOSPLIT, True, [])]
    qkv = [n * H, g * H, g * H]
    p = "decoder.layer.0."

    def lowp(name, K, N, split, groups=()):
        recs.append((name + ".weight", rng.integers(0, 256, (K, N // 2), dtype=np.uint8), split, False, [x // 2 for x in groups]))
        recs.append((name + ".weight.scale", bf((K // G, N), 0.01), split, True, list(groups)))
        recs.append((name + ".weight.zero_point", bf((K // G, N), 3.0), split, True, list(groups)))

    recs.append((p + "attention.layernorm.gamma", bf((hidden,), 1.0), NOSPLIT, True, []))
    lowp(p + "attention.self", hidden, sum(qkv), GROUP_VSPLIT, qkv)
    recs.append((p + "attention.self.bias", bf((sum(qkv),), 0.1), GROUP_VSPLIT, True, qkv))
    lowp(p + "attention.output.dense", n * H, hidden, HSPLIT)
    recs.append((p + "attention.output.dense.bias", bf((hidden,), 0.1), HSPLIT, True, []))
    recs.append((p + "ffn.layernorm.gamma", bf((hidden,), 1.0), NOSPLIT, True, []))
    lowp(p + "ffn.intermediate.dense", hidden, inter, VSPLIT)
    lowp(p + "ffn.linear.dense", hidden, inter, VSPLIT)
    lowp(p + "ffn.output.dense", inter, hidden, HSPLIT)
    # a per-channel int8 row-parallel weight: the parameters are whole on every rank (qwen_v15.py:556-569)
    recs.append(("perchannel.o.weight", rng.integers(-128, 128, (n * H, hidden), dtype=np.int8), HSPLIT, False, []))
    recs.append(("perchannel.o.weight.scale", bf((hidden,), 0.01), NOSPLIT, True, []))
    recs.append(("perchannel.o.weight.zero_point", bf((hidden,), 3.0), NOSPLIT, True, []))
    recs.append(("final.layernorm.gamma", bf((hidden,), 1.0), NOSPLIT, True, []))
    recs.append(("lm_head.weight", bf((hidden, vocab)), VSPLIT, True, []))
    # one tensor per remaining splitter
    recs.append(("qkvsplit.weight", bf((8, 3 * 64)), QKVSPLIT, True, []))
    recs.append(("mqa.weight", bf((8, 256 + 32 + 32)), MQA_VSPLIT, True, [256, 32, 32]))
    recs.append(("experts.gate_up.weight", rng.integers(-128, 128, (4, 16, 64), dtype=np.int8), BATCH_VSPLIT, False, []))
    recs.append(("experts.gate_up.scale", bf((4, 64), 0.01), BATCH_VSPLIT, True, []))
    recs.append(("experts.down.weight", rng.integers(-128, 128, (4, 32, 24), dtype=np.int8), BATCH_HSPLIT, False, []))
    recs.append(("experts.ep.weight", rng.integers(-128, 128, (8, 6, 10), dtype=np.int8), EPSPLIT, False, []))
    return recs


def tiny_model(seed=5):
    """1 layer, hidden 256, 2 query / 1 KV head of 128, intermediate 512, vocabulary 320, int4 g128: [(name, array, split, bf16)]"""
    rng = np.random.default_rng(seed)
    hidden, n, g, H, inter, vocab, G = 256, 2, 1, 128, 512, 320, 128
    bf = lambda shape, s=0.05: (rng.normal(0, s, shape).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)   # bf16 bit patterns (truncated)
    recs = [("embedding.word_embeddings", bf((vocab, hidden)), NOSPLIT, True)]

    def lowp(name, K, N, split):
        recs.append((name + ".weight", rng.integers(0, 256, (K, N // 2), dtype=np.uint8), split, False))                # two nibbles per byte
        recs.append((name + ".weight.scale", bf((K // G, N), 0.01), split, True))
        recs.append((name + ".weight.zero_point", bf((K // G, N), 3.0), split, True))

    p = "decoder.layer.0."
    recs.append((p + "attention.layernorm.gamma", bf((hidden,), 1.0), NOSPLIT, True))
    lowp(p + "attention.self", hidden, (n + 2 * g) * H, GROUP_VSPLIT)
    recs.append((p + "attention.self.bias", bf(((n + 2 * g) * H,), 0.1), GROUP_VSPLIT, True))
    lowp(p + "attention.output.dense", n * H, hidden, HSPLIT)
    recs.append((p + "ffn.layernorm.gamma", bf((hidden,), 1.0), NOSPLIT, True))
    lowp(p + "ffn.intermediate.dense", hidden, inter, VSPLIT)
    lowp(p + "ffn.linear.dense", hidden, inter, VSPLIT)
    lowp(p + "ffn.output.dense", inter, hidden, HSPLIT)
    recs.append(("final.layernorm.gamma", bf((hidden,), 1.0), NOSPLIT, True))
    recs.append(("lm_head.weight", bf((hidden, vocab)), VSPLIT, True))
    recs.append(("a.scalar.f32", np.array([1.5], np.float32), NOSPLIT, False))          # rank-1 shapes print as "(1,)"
    recs.append(("an.int64.table", np.arange(12, dtype=np.int64).reshape(3, 4), NOSPLIT, False))
    return recs


def sparse_model(seed=3):
    """Two 2-D f32 matrices for the writer's sparse encodings (CSC / ELL): ~85 % zeros, an empty column, a full column, every non-zero well above
    the writers' 1e-9 threshold; and one dense record behind them (the index must step over the compressed data).  [(name, array, split, bf16, enc)]"""
    rng = np.random.default_rng(seed)
    def mat(rows, cols):
        a = rng.normal(0, 1, (rows, cols)).astype(np.float32)
        a[np.abs(a) < 1e-3] = 0.5
        a[rng.random((rows, cols)) < 0.85] = 0.0
        a[:, 3] = 0.0
        a[:, 5] = rng.normal(0, 1, rows).astype(np.float32) + 3.0
        return a
    return [("sparse.csc", mat(48, 24), NOSPLIT, False, "csc"), ("sparse.ell", mat(40, 16), NOSPLIT, False, "ell"),
            ("sparse.csc.vsplit", mat(16, 32), VSPLIT, False, "csc"), ("dense.after", np.arange(10, dtype=np.float32).reshape(2, 5), NOSPLIT, False)]


def main_sparse():
    path = os.path.join(OUT, "tiny_sparse.asparam")
    write(path, sparse_model())
    print("wrote", path, os.path.getsize(path), "bytes")


def main_tp():
    path = os.path.join(OUT, "tiny_qwen2_a16w4_tp.asparam")
    write(path, tp_model())
    print("wrote", path, os.path.getsize(path), "bytes")


def main():
    recs = tiny_model()
    path = os.path.join(OUT, "tiny_qwen2_a16w4.asparam")
    write(path, recs)
    print(path, os.path.getsize(path), "bytes,", len(recs), "records")


if __name__ == "__main__":
    main()
    main_tp()
    main_sparse()
