"""Launch-plan heuristics that need no GPU (host code of libdashinfer_hip.so): the split plan of the decode attention.

The numbers are the ones the round-3 measurements settled on (profiles/r03_attn_timeline.txt, r03ab_*): 17 splits at batch 1,
4 at BASELINE configs[2] (batch 32, uint4 cache), 16 -- not 32 -- for one TP = 8 rank of Qwen2-72B, 4 -- not 8 -- for the 57B MoE
step; and the invariants behind them."""
import ctypes as C
import importlib

import pytest

capi = importlib.import_module("dash-infer_amd.capi")


def plan(batch, n_heads, n_groups, max_len, kv="none", dtype=None, cus=256):
    lib = capi.lib()
    ns, mf = C.c_int(), C.c_int()
    rc = lib.dihip_debug_attn_plan(batch, n_heads, n_groups, max_len, capi.KV[kv],
                                   capi.BF16 if dtype is None else dtype, cus, C.byref(ns), C.byref(mf))
    assert rc == 0
    return ns.value, bool(mf.value)


@pytest.mark.parametrize("shape,expect", [
    ((1, 28, 4, 2049, "none"), 17),      # headline: 17 x 4 workgroups of 128 tokens
    ((32, 28, 4, 2049, "u4"), 4),        # configs[2]: two workgroups per CU, 512 tokens per split
    ((16, 8, 1, 4097, "none"), 16),      # configs[3] rank: one workgroup per CU (32 splits of 128 tokens measured 3 us slower)
    ((16, 28, 4, 1025, "none"), 4),      # configs[4] step
])
def test_measured_shapes_keep_their_split_counts(shape, expect):
    b, n, g, L, kv = shape
    ns, mfma = plan(b, n, g, L, kv)
    assert mfma and ns == expect


@pytest.mark.parametrize("batch", [1, 2, 3, 8, 16, 32, 64, 200])
@pytest.mark.parametrize("n,g", [(28, 4), (8, 1), (64, 8), (32, 32)])
@pytest.mark.parametrize("L", [1, 100, 129, 1024, 2049, 4097, 32768])
def test_split_plan_invariants(batch, n, g, L):
    ns, _ = plan(batch, n, g, L)
    assert 1 <= ns <= 256
    assert ns == 1 or (L + ns - 1) // ns >= 64           # a split never falls under half the 128-token floor's rounding
    assert ns <= max(1, (L + 127) // 128)                # >= 128 tokens per split before rounding
    if batch * g * ns > 256 + batch * g:                 # clearly more than one workgroup per CU (past the rounding of the split
        assert ns == 1 or (L + ns - 1) // ns >= 256      # count) only while a split keeps >= 256 tokens


def test_plan_rejects_bad_shapes():
    lib = capi.lib()
    assert lib.dihip_debug_attn_plan(1, 28, 5, 128, 0, capi.BF16, 256, None, None) != 0
    assert lib.dihip_debug_attn_plan(0, 28, 4, 128, 0, capi.BF16, 256, None, None) != 0
