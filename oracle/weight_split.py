"""oracle/weight_split.py -- TEST INFRASTRUCTURE (only tests/ import it): numpy restatement of the reference's tensor-parallel weight
splitters, csrc/runtime/weight/weight_splitter.cpp, as WeightManager applies them per rank when a model is loaded
(GetSplitterByMode :921-960).  `a` is the whole tensor as the converter stored it; returns the rank's share.

Parity: the reference's splitter classes sit on AsTensor / TensorUtils / glog and cannot be compiled from where they lie without
stand-ins for those (unbuildable here: "parity unpinned" against the reference's binary); the restatement follows the cited lines,
and is cross-checked against dash-infer_amd/tp.py (round 2's independent statement of the same rules) in tests/test_host_weight_split.py."""
import numpy as np

NOSPLIT, VSPLIT, HSPLIT, QKVSPLIT, KVSPLIT, HSPLIT_QUANTIZE, GROUP_VSPLIT, MQA_VSPLIT, BATCH_VSPLIT, BATCH_HSPLIT, BATCH_KVSPLIT, EPSPLIT = range(12)


class NotSplittable(ValueError):
    """IsSplittable() == false (the reference logs an error and fails the load)"""


def _div(x, r, what):
    if x % r:
        raise NotSplittable(f"{what}: {x} does not divide by {r} ranks")
    return x // r


def split(a, mode, rank, nranks, group_list=()):
    a = np.asarray(a)
    if nranks <= 1 or mode == NOSPLIT:                       # WeightSplitterNoSplit :22-48
        return a.copy()
    if mode == VSPLIT:                                        # :60-127: columns of a matrix, the same range of a bias
        if a.ndim not in (1, 2):
            raise NotSplittable("VSPLIT of a higher-rank tensor")
        w = _div(a.shape[-1], nranks, "VSPLIT")
        return a[..., rank * w:(rank + 1) * w].copy()
    if mode == HSPLIT:                                        # :369-438
        if a.ndim == 1:                                       # :424-432 the bias is kept on rank 0, zero elsewhere (added once)
            return a.copy() if rank == 0 else np.zeros_like(a)
        if a.ndim != 2:
            raise NotSplittable("HSPLIT of a higher-rank tensor")
        h = _div(a.shape[0], nranks, "HSPLIT")
        return a[rank * h:(rank + 1) * h].copy()
    if mode in (QKVSPLIT, KVSPLIT):                           # WeightSplitterVSplitBatchGEMM<3 | 2> :521-610
        cnt = 3 if mode == QKVSPLIT else 2
        if a.ndim not in (1, 2) or a.shape[-1] % (cnt * nranks):
            raise NotSplittable("QKVSPLIT / KVSPLIT")
        g = a.shape[-1] // cnt
        return np.concatenate([a[..., i * g + rank * (g // nranks): i * g + (rank + 1) * (g // nranks)] for i in range(cnt)], -1)
    if mode == GROUP_VSPLIT:                                  # :611-721: 1/R of every group, concatenated
        if a.ndim not in (1, 2) or not group_list or sum(group_list) != a.shape[-1]:
            raise NotSplittable("GROUP_VSPLIT: group_list")
        parts, at = [], 0
        for g in group_list:
            w = _div(g, nranks, "GROUP_VSPLIT group")
            parts.append(a[..., at + rank * w: at + (rank + 1) * w])
            at += g
        return np.concatenate(parts, -1)
    if mode == MQA_VSPLIT:                                    # :722-852: q split, the K / V head whole on every rank
        if a.ndim not in (1, 2) or len(group_list) != 3 or sum(group_list) != a.shape[-1]:
            raise NotSplittable("MQA_VSPLIT: group_list")
        q = _div(group_list[0], nranks, "MQA_VSPLIT q")
        return np.concatenate([a[..., rank * q:(rank + 1) * q], a[..., group_list[0]:]], -1)
    if mode == BATCH_VSPLIT:                                  # :128-232 ([E, K, N] and the [E, N] parameters: CopyWeight handles both)
        if a.ndim not in (2, 3):
            raise NotSplittable("BATCH_VSPLIT of a vector")
        w = _div(a.shape[-1], nranks, "BATCH_VSPLIT")
        return a[..., rank * w:(rank + 1) * w].copy()
    if mode == BATCH_HSPLIT:                                  # :439-520: rank-3 only
        if a.ndim != 3:
            raise NotSplittable("BATCH_HSPLIT of a tensor that is not [E, K, N]")
        h = _div(a.shape[1], nranks, "BATCH_HSPLIT")
        return a[:, rank * h:(rank + 1) * h].copy()
    if mode == EPSPLIT:                                       # :853-919
        if a.ndim != 3:
            raise NotSplittable("EPSPLIT of a tensor that is not [E, K, N]")
        e = _div(a.shape[0], nranks, "EPSPLIT")
        return a[rank * e:(rank + 1) * e].copy()
    raise NotSplittable(f"no splitter for SplitMode {mode} (GetSplitterByMode :921-960)")
