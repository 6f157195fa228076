"""Tensor-parallel partitioning of a Qwen2-style decoder layer (host logic, torch CPU/GPU agnostic).

Mirrors the reference's weight splitters (csrc/runtime/weight/weight_splitter.cpp):
  qkv      GROUP_VSPLIT  [nH, gH, gH] each divided over ranks          (:611-721, qwen_v15.py:130-137)
  o_proj   HSPLIT        rows (K) follow the rank's query heads         (:369-438)
  gate/up  VSPLIT        columns                                        (:60-127)
  down     HSPLIT        rows follow the rank's gate/up columns
  per-channel scales/zeros of HSPLIT weights are NOT split, sub-channel ones are split along G
  (qwen_v15.py:540-569), which requires each rank's K range to be whole quantisation groups.
New relative to the reference: when there are fewer KV heads than ranks
(Qwen2-7B: g = 4, TP = 8) each KV head is REPLICATED on nranks/g ranks and its query heads are
divided (unevenly if necessary: 7 = 4 + 3) between them; the reference throws PARAM_ERROR
(head_gqa.h:29-49).  FFN columns are divided in units of the quantisation group so that the
down-projection's K split stays group aligned (18944/8 = 2368 is not a multiple of 128:
ranks get 19 or 18 groups).
"""
from dataclasses import dataclass
from typing import List


def split_units(total_units: int, parts: int) -> List[int]:
    """Divide `total_units` into `parts` contiguous chunks whose sizes differ by at most 1
    (larger chunks first)."""
    base, rem = divmod(total_units, parts)
    return [base + (1 if i < rem else 0) for i in range(parts)]


@dataclass
class HeadShard:
    q_heads: List[int]   # global query-head indices owned by the rank
    kv_heads: List[int]  # global KV-head indices held (possibly replicated) by the rank


def shard_heads(n: int, g: int, nranks: int) -> List[HeadShard]:
    hpg = n // g
    assert n % g == 0
    out = []
    if g >= nranks:
        if g % nranks != 0:
            raise ValueError(f"num KV heads {g} not divisible by nranks {nranks}")  # head_gqa.h:29-49
        per = g // nranks
        for r in range(nranks):
            kv = list(range(r * per, (r + 1) * per))
            out.append(HeadShard([h for k in kv for h in range(k * hpg, (k + 1) * hpg)], kv))
    else:
        if nranks % g != 0:
            raise ValueError(f"nranks {nranks} not a multiple of num KV heads {g}")
        rep = nranks // g  # ranks sharing one KV head
        for r in range(nranks):
            k = r // rep
            sizes = split_units(hpg, rep)
            start = k * hpg + sum(sizes[: r % rep])
            out.append(HeadShard(list(range(start, start + sizes[r % rep])), [k]))
    return out


def shard_ffn(inter: int, nranks: int, unit: int) -> List[range]:
    """Column ranges of gate/up (= row ranges of down) per rank, in multiples of `unit`."""
    if inter % unit != 0:
        raise ValueError(f"intermediate size {inter} is not a multiple of the quantisation group {unit} "
                         "(qwen_v15.py:540-547)")
    sizes = split_units(inter // unit, nranks)
    out, pos = [], 0
    for s in sizes:
        out.append(range(pos * unit, (pos + s) * unit))
        pos += s
    return out


def qkv_columns(shard: HeadShard, n: int, g: int, H: int) -> List[int]:
    """Columns of the fused [K, (n+2g)H] qkv weight (and bias) owned by a rank, in the order
    [its q heads | its k heads | its v heads]."""
    cols = []
    for h in shard.q_heads:
        cols.extend(range(h * H, (h + 1) * H))
    for k in shard.kv_heads:
        cols.extend(range((n + k) * H, (n + k + 1) * H))
    for k in shard.kv_heads:
        cols.extend(range((n + g + k) * H, (n + g + k + 1) * H))
    return cols


def o_rows(shard: HeadShard, H: int) -> List[int]:
    rows = []
    for h in shard.q_heads:
        rows.extend(range(h * H, (h + 1) * H))
    return rows
