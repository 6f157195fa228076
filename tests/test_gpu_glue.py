"""GPU parity of the glue ops (RMSNorm, RoPE, Binary ADD, SiLU*MUL, embedding, step counter)
against oracle/glue.py (which restates csrc/core/kernel/cpu/layernorm.cpp:110-157 and
rotary.cpp:22-106)."""
import numpy as np
import pytest
import torch

from oracle import glue
from oracle.numerics import bf16_round

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops(pkg):
    from dash_infer_amd import ops as _ops
    assert torch.cuda.is_available()
    return _ops


def dev(a, dtype=torch.bfloat16):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dtype).cuda()


@pytest.mark.parametrize("rows,cols", [(1, 3584), (5, 896), (32, 1000)])
def test_rmsnorm(ops, rows, cols):
    rng = np.random.default_rng(rows)
    x = bf16_round(rng.normal(0, 2, (rows, cols)).astype(np.float32))
    gamma = bf16_round(rng.normal(1, 0.2, cols).astype(np.float32))
    y = ops.rmsnorm(dev(x), dev(gamma), 1e-6).float().cpu().numpy()
    ref = bf16_round(glue.rmsnorm(x, gamma, 1e-6))
    np.testing.assert_allclose(y, ref, rtol=2 ** -7, atol=1e-6)
    xf = rng.normal(0, 2, (rows, cols)).astype(np.float32)
    yf = ops.rmsnorm(dev(xf, torch.float32), dev(gamma, torch.float32), 1e-6).cpu().numpy()
    np.testing.assert_allclose(yf, glue.rmsnorm(xf, gamma, 1e-6), rtol=1e-5, atol=1e-6)


def test_rope_qk(ops):
    rng = np.random.default_rng(2)
    n, g, H, rows = 14, 2, 128, 6
    inv = glue.rope_inv_freq(H, 1000000.0)
    pos = np.array([0, 1, 17, 2047, 4095, 33], np.int32)
    qkv = rng.normal(0, 1, (rows, (n + 2 * g) * H)).astype(np.float32)
    t = dev(qkv, torch.float32)
    ops.rope_qk_(t, torch.from_numpy(pos).cuda(), torch.from_numpy(inv).cuda(), n, g, H)
    out = t.cpu().numpy()
    ref = qkv.copy()
    heads = qkv[:, : (n + g) * H].reshape(rows, n + g, H)
    ref[:, : (n + g) * H] = glue.rope(heads, pos, inv).reshape(rows, -1)
    np.testing.assert_allclose(out, ref, rtol=1e-4, atol=2e-4)  # sincosf of angles up to 4095 rad
    np.testing.assert_array_equal(out[:, (n + g) * H:], qkv[:, (n + g) * H:])  # V untouched


def test_binary_ops_and_embedding(ops):
    rng = np.random.default_rng(3)
    a = bf16_round(rng.normal(0, 2, 100003).astype(np.float32))
    b = bf16_round(rng.normal(0, 2, 100003).astype(np.float32))
    np.testing.assert_array_equal(ops.binary_add(dev(a), dev(b)).float().cpu().numpy(), bf16_round(a + b))
    sm = ops.silu_mul(dev(a), dev(b)).float().cpu().numpy()
    ref = bf16_round(bf16_round(glue.silu(a)) * b)
    np.testing.assert_allclose(sm, ref, rtol=2 ** -6, atol=1e-6)
    table = bf16_round(rng.normal(0, 1, (50, 96)).astype(np.float32))
    ids = torch.tensor([3, 49, 0, 3], dtype=torch.int64, device="cuda")
    np.testing.assert_array_equal(ops.embedding(ids, dev(table)).cpu().numpy(), table[[3, 49, 0, 3]])
    v = torch.tensor([0, 5, 2047], dtype=torch.int32, device="cuda")
    assert ops.increment_u32_(v).tolist() == [1, 6, 2048]


def test_nan_logits_and_wild_ids_stay_in_range(ops):
    """an all-NaN logits row must give a valid token id (0), not the 2^31 - 1 sentinel, and an id out of
    range must not make the embedding read outside its table."""
    logits = torch.randn(3, 5000, device="cuda")
    logits[1] = float("nan")
    ids = ops.argmax(logits)
    torch.cuda.synchronize()
    assert ids.tolist()[0] == int(torch.argmax(logits[0])) and ids.tolist()[1] == 0 and ids.tolist()[2] == int(torch.argmax(logits[2]))
    table = torch.randn(10, 64, device="cuda").to(torch.bfloat16)
    wild = torch.tensor([3, 2 ** 31 - 1, -5, 9], dtype=torch.int64, device="cuda")
    h = ops.embedding(wild, table)
    torch.cuda.synchronize()
    assert torch.equal(h[0], table[3].float()) and torch.equal(h[1], table[9].float()) and torch.equal(h[2], table[0].float())
