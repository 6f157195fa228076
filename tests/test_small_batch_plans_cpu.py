"""Host-side plans of round 5 that need no GPU (the C-ABI library answers them from shapes alone; without a device the CU count
defaults to 256, the MI355X's):

  * dihip_gemm_prefill_tail_parts -- the context-phase GEMM's tail split (gemm_prefill_kernel.hpp, PrefillArgs::tail_cb): the grids
    DESIGN.md section 3.5 quotes for Qwen2-7B at 2048 rows, and the shapes that must NOT be split;
  * dihip_prenorm_rowsq_supported -- which consumers take a deferred RMSNorm (include/dashinfer_hip.h section 1).
Reference shapes: Qwen2-7B (hidden 3584, 28 / 4 heads, intermediate 18944) and one TP = 8 rank of Qwen2-72B (hidden 8192, intermediate 3712)."""
import pytest

BF16, F16 = 2, 1          # capi.BF16 / capi.F16 are read from the package below; kept here for the table's readability
ROW, FRAG = 0, 1


@pytest.fixture(scope="module")
def lib(pkg):
    from dash_infer_amd import capi
    return capi.lib()


def test_prefill_tail_split_plans(lib):
    parts = lambda wbits, M, N, K, G, dual: int(lib.dihip_gemm_prefill_tail_parts(wbits, M, N, K, G, dual))
    # Qwen2-7B int4 g128, one 2048-token prompt: qkv 18 x 16 = 288 tiles -> the last 2 column blocks in 7 parts; gate / up 148 x 16 = 2368 -> 4 parts
    assert parts(4, 2048, 4608, 3584, 128, 0) == 7
    assert parts(4, 2048, 18944, 3584, 128, 1) == 4
    # o / down: 224 tiles, one round at 87.5 %: nothing to split; a grid that divides the chip: nothing either
    assert parts(4, 2048, 3584, 3584, 128, 0) == 1
    assert parts(4, 2048, 3584, 18944, 128, 0) == 1
    assert parts(4, 2048, 4096, 3584, 128, 0) == 1
    # a last round more than half full stays a round of tiles; short prompts (one round or less) too; decode batches never
    assert parts(4, 2048, 6400, 3584, 128, 0) == 1      # 25 x 16 = 400 tiles: 144 in the second round
    assert parts(4, 512, 4608, 3584, 128, 0) == 1       # 18 x 4 = 72 tiles
    assert parts(4, 32, 4608, 3584, 128, 0) == 1
    # parts hold whole quantisation groups: g256 over K = 1024 has 4 groups, per-channel int8 splits in k-tiles of 64
    assert parts(4, 2048, 4608, 1024, 256, 0) == 4
    assert parts(8, 2048, 4608, 512, -1, 0) == 8
    # the slab is covered by the workspace the callers size with dihip_gemm_lowp_workspace_bytes
    need = 32 * 7 * 128 * 256 * 4
    assert int(lib.dihip_gemm_lowp_workspace_bytes(4, 2048, 4608, 3584, 128)) >= need + 2048 * 3584 * 2


def test_deferred_norm_consumers(lib, pkg):
    from dash_infer_amd import capi
    sup = lambda wbits, M, N, K, G, dual, dt, lay: bool(lib.dihip_prenorm_rowsq_supported(wbits, M, N, K, G, dual, dt, lay))
    bf16, f16 = capi.BF16, capi.F16
    # the gate / up pair of Qwen2-7B on the K-slice kernel (FRAG32 activations), batch 5 .. 32; of the 72B rank (split-K slab: the reduction applies 1 / rms)
    for M in (5, 16, 17, 32):
        assert sup(4, M, 18944, 3584, 128, 1, bf16, FRAG)
        assert sup(4, M, 3712, 8192, 128, 1, bf16, FRAG)
    # the qkv projection on the whole-column kernel: plain epilogue, either layout
    assert sup(4, 32, 4608, 3584, 128, 0, bf16, ROW) and sup(4, 9, 4608, 3584, 128, 0, bf16, FRAG)
    # not: f16 (un-normalised gamma * h may leave the f16 range), batch 1 .. 4 (the GEMV normalises in its prologue), the context phase,
    # a SwiGLU pair that the whole-column kernel would serve (row-major activations)
    assert not sup(4, 32, 18944, 3584, 128, 1, f16, FRAG)
    assert not sup(4, 4, 18944, 3584, 128, 1, bf16, FRAG) and not sup(4, 1, 4608, 3584, 128, 0, bf16, ROW)
    assert not sup(4, 64, 18944, 3584, 128, 1, bf16, FRAG)
    assert not sup(4, 16, 1024, 512, 128, 1, bf16, ROW)
    assert int(lib.dihip_rowsq_bytes()) == 256 * 32 * 4
