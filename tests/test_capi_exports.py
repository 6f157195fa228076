"""The C-ABI library loads and exports every symbol that include/dashinfer_hip.h declares
(no compute calls: runs without a GPU), the ctypes table covers the whole header, and the
size-query entry points behave on the CPU."""
import ctypes
import os

import pytest


def test_library_exports_every_header_symbol(pkg):
    from dash_infer_amd import capi
    assert os.path.exists(pkg.LIB_PATH), "build with __graft_entry__.build()"
    l = ctypes.CDLL(pkg.LIB_PATH)
    syms = capi.header_symbols()
    assert len(syms) >= 40
    missing = [s for s in syms if not hasattr(l, s)]
    assert not missing, f"declared in the header but not exported: {missing}"
    assert set(syms) == set(capi._SIGS), set(syms) ^ set(capi._SIGS)


def test_size_queries_without_gpu(pkg):
    from dash_infer_amd import capi
    l = capi.lib()
    assert b"gfx950" in l.dihip_version()
    # Qwen2-7B o_proj, int4: K=N=3584 -> 224 n-tiles x 28 k-tiles x 1 KiB
    assert l.dihip_gemm_lowp_packed_weight_bytes(4, 3584, 3584) == 224 * 28 * 1024
    assert l.dihip_gemm_lowp_packed_weight_bytes(8, 3584, 3584) == 224 * 56 * 1024
    assert l.dihip_gemm_lowp_packed_weight_bytes(3, 3584, 3584) == 0
    assert l.dihip_gemm_lowp_packed_sz_bytes(3584, 3584, 128) == 28 * 3584 * 4
    assert l.dihip_gemm_lowp_packed_sz_bytes(3584, 3584, -1) == 3584 * 4
    # span bytes = CacheUtils::GetSpanSizeInBytes (virtual_cache.cpp:202-232)
    assert l.dihip_span_bytes(4, 128, 128, capi.KV_NONE, capi.BF16) == 4 * 128 * 128 * 2
    assert l.dihip_span_bytes(4, 128, 128, capi.KV_I8, capi.BF16) == 4 * 128 * 128 + 2 * 128 * 4 * 4
    assert l.dihip_span_bytes(4, 128, 128, capi.KV_U4, capi.BF16) == 4 * 128 * 64 + 2 * 128 * 4 * 4
    assert l.dihip_gemm_lowp_sync_bytes() > 0


def test_span_attn_handle_error_behaviour_matches_reference(pkg):
    """span-attention/test/test_lib/test_api.cpp:345-413 (create-time failure cases; they do
    not touch the device)."""
    from dash_infer_amd import capi
    l = capi.lib()

    def create(dtype, mode, batch, seqlen, heads, groups, hs, span, maxlen):
        h = ctypes.c_void_p()
        lens = (ctypes.c_int * max(batch, 1))(*([seqlen] * max(batch, 1)))
        nspans = (maxlen + span - 1) // span if span > 0 else 0
        st = l.dihip_span_attn_create_handle(ctypes.byref(h), dtype, mode, batch, heads, groups, hs, span, nspans, lens, 256)
        return st, h

    F32, NONE = capi.F32, capi.KV_NONE
    assert create(F32, NONE, 14, 181239920, 16, 2, 128, 16, 181239922)[0] == capi.SA_EXCEED_LIMIT_ERROR
    assert create(F32, NONE, 14, 1024, 15, 2, 128, 16, 1024)[0] == capi.SA_PARAM_ERROR   # heads % groups
    assert create(F32, NONE, 14, 1024, -8, 2, 128, 512, 1024)[0] == capi.SA_PARAM_ERROR
    assert create(F32, NONE, 14, 1024, 8, -1, 128, 512, 1024)[0] == capi.SA_PARAM_ERROR
    assert create(F32, NONE, 14, 1024, 8, 2, 512, 512, 1024)[0] == capi.SA_PARAM_ERROR   # head size
    assert create(F32, NONE, 0, 1024, 8, 2, 128, 32, 1024)[0] == capi.SA_PARAM_ERROR
    assert create(F32, NONE, 14, 1024, 0, 2, 128, 32, 1024)[0] == capi.SA_PARAM_ERROR
    assert create(F32, NONE, 14, 1024, 8, 0, 128, 32, 1024)[0] == capi.SA_PARAM_ERROR
    assert create(F32, NONE, 14, 1024, 8, 2, 0, 32, 1024)[0] == capi.SA_PARAM_ERROR
    # success cases of test_api.cpp:278-343 (create / workspace query / destroy)
    for dtype in (capi.F32, capi.F16, capi.BF16):
        for mode in (capi.KV_NONE, capi.KV_I8, capi.KV_U4):
            st, h = create(dtype, mode, 14, 999, 16, 2, 128, 16, 999)
            assert st == capi.SA_SUCCESS and h.value
            dws, hws = ctypes.c_size_t(), ctypes.c_size_t()
            assert l.dihip_span_attn_device_workspace_bytes(ctypes.byref(dws), h) == 0 and dws.value > 0
            assert l.dihip_span_attn_host_workspace_bytes(ctypes.byref(hws), h) == 0
            assert l.dihip_span_attn_destroy_handle(h) == 0
    # run-time failures: created fine, Run rejects (test_api.cpp:394-413); fake pointers never touched
    for heads, groups, span in ((68, 2, 16), (8, 2, 512)):
        st, h = create(F32, NONE, 14, 1024, heads, groups, 128, span, 1024)
        assert st == capi.SA_SUCCESS
        fake = ctypes.c_void_p(0xdeadbeef)
        assert l.dihip_span_attn_run(fake, fake, fake, fake, fake, 1 << 30, fake, 1 << 20, 0.1, h, None) == capi.SA_PARAM_ERROR
        l.dihip_span_attn_destroy_handle(h)
    st, h = create(F32, NONE, 14, 1024, 8, 2, 128, 16, 1024)
    fake = ctypes.c_void_p(0xdeadbeef)
    assert l.dihip_span_attn_run(None, None, fake, fake, fake, 1 << 30, fake, 1 << 20, 0.1, h, None) == capi.SA_PARAM_ERROR
    assert l.dihip_span_attn_run(fake, fake, fake, fake, None, 1 << 30, None, 1 << 20, 0.1, h, None) == capi.SA_PARAM_ERROR
    l.dihip_span_attn_destroy_handle(h)
    assert l.dihip_span_attn_destroy_handle(None) == capi.SA_PARAM_ERROR
