"""GenerateOp's logits processors pinned against the REFERENCE'S OWN DEVICE CODE: oracle/_ref/libdashinfer_ref_logits.so is
cuda::LogitsProcessor<float> and its five kernels (csrc/core/kernel/cuda/beam_search.cu:329-539) with BatchGencfg (csrc/common/common.h:271-282),
sliced from where they lie and compiled for gfx950 (oracle/logits_ref.hip, oracle/Makefile `reflogits`; the CUDA runtime calls spelled as their HIP
equivalents) -- six launches, a copy of the scores, a memset of the [batch, vocab] count array, as the reference runs them.

The product's ONE token-driven launch (dihip_logits_processor) and the numpy restatement (oracle/logits_proc.py) both equal the reference's kernels
BIT FOR BIT (built without contraction: `count * frequency` and `+ presence` are two roundings; a fused multiply-add, which nvcc may pick on the
reference's own platform, could move a penalised logit by one rounding of the penalty).
Inputs stay inside what the reference defines: ids inside the vocabulary (its n-gram kernel indexes the scores unchecked), cur_len <= max_len."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import logits_proc as lp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ref():
    path = os.path.join(ROOT, "oracle", "_ref", "libdashinfer_ref_logits.so")
    if not os.path.exists(path):
        pytest.skip(f"{path} not built (oracle/Makefile reflogits needs /root/reference; the prebuilt library travels with gpurun)")
    lib = C.CDLL(path)
    lib.ref_logits_processor.restype = C.c_int
    lib.ref_logits_processor.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 10 + [C.c_size_t, C.c_void_p]
    return lib


def _run_ref(lib, logits, ids, cur, inp, rep, freq, pres, ng, minl, eos, sup):
    M, N = logits.shape
    d = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).cuda()
    x = d(logits, np.float32)
    t = dict(ids=d(ids, np.int64), rep=d(rep, np.float32), pres=d(pres, np.float32), freq=d(freq, np.float32), ng=d(ng, np.int32), minl=d(minl, np.int32),
             eos=d(eos, np.int32), cur=d(cur, np.int32), inp=d(inp, np.int32), sup=d(sup, np.int32))
    ws = torch.empty(M * N * 4, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    rc = lib.ref_logits_processor(x.data_ptr(), t["ids"].data_ptr(), M, ids.shape[1], N, t["rep"].data_ptr(), t["pres"].data_ptr(), t["freq"].data_ptr(),
                                  t["ng"].data_ptr(), t["minl"].data_ptr(), t["eos"].data_ptr(), t["cur"].data_ptr(), t["inp"].data_ptr(),
                                  t["sup"].data_ptr(), ws.data_ptr(), ws.numel(), None)
    assert rc == 0
    torch.cuda.synchronize()
    return x.cpu().numpy()


def _case(rng, M, N, max_len):
    ids = rng.integers(0, N, (M, max_len)).astype(np.int64)
    hot = rng.integers(0, N, 10)
    for b in range(M):
        mask = rng.random(max_len) < 0.6
        ids[b, mask] = hot[rng.integers(0, len(hot), int(mask.sum()))]
    cur = rng.integers(1, max_len + 1, M).astype(np.int32)
    inp = np.minimum(cur, rng.integers(0, max_len // 2 + 1, M)).astype(np.int32)
    cur[0], inp[0] = max_len, 0
    return dict(ids=ids, cur=cur, inp=inp, rep=rng.choice([1.0, 1.1, 1.3, 0.8], M).astype(np.float32),
                freq=rng.choice([0.0, 0.1, 0.37, -0.2], M).astype(np.float32), pres=rng.choice([0.0, 0.5, 1.2], M).astype(np.float32),
                ng=rng.choice([0, 2, 3, 1, 4], M).astype(np.int32), minl=rng.integers(0, max_len + 10, M).astype(np.int32),
                eos=rng.integers(0, N, M).astype(np.int32), sup=rng.integers(0, 2, M).astype(np.int32))


@pytest.mark.parametrize("M,N,max_len", [(1, 152064, 2048), (8, 32000, 300), (32, 4096, 64), (4, 50, 40)])
def test_product_and_oracle_equal_the_reference_kernels_bit_for_bit(pkg, M, N, max_len):
    from dash_infer_amd import ops
    lib = _ref()
    rng = np.random.default_rng(M * 7 + max_len)
    c = _case(rng, M, N, max_len)
    logits = rng.normal(0, 4, (M, N)).astype(np.float32)
    args = (c["ids"], c["cur"], c["inp"], c["rep"], c["freq"], c["pres"], c["ng"], c["minl"], c["eos"], c["sup"])
    ref = _run_ref(lib, logits, *args)
    want = lp.logits_processor(logits, *args)
    x = torch.from_numpy(logits).cuda()
    ops.logits_processor_(x, torch.from_numpy(c["ids"]).cuda(), *args[1:])
    torch.cuda.synchronize()
    got = x.cpu().numpy()
    assert (ref != logits).any()
    assert (want.view(np.uint32) == ref.view(np.uint32)).all(), f"oracle vs the reference's kernels: {int((want != ref).sum())} logits differ"
    assert (got.view(np.uint32) == ref.view(np.uint32)).all(), f"product vs the reference's kernels: {int((got != ref).sum())} logits differ"
