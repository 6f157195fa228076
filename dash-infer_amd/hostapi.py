"""ctypes binding of include/dashinfer_hip_host.h: the C test harness around the C++ operator
layer (libdashinfer_hip_ops.so).  Plumbing for tests; fails loudly when the library is missing."""
import ctypes as C
import os
import re

from . import REPO_ROOT
from .capi import lib as _devlib

OPS_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libdashinfer_hip_ops.so")
DT = {"f32": 1, "f16": 2, "i8": 3, "i32": 5, "i64": 6, "bf16": 9, "u8": 10}
vp, i32, sz = C.c_void_p, C.c_int, C.c_size_t
_SIGS = {
    "dihost_model_create": (i32, [C.POINTER(vp), vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "dihost_model_destroy": (i32, [vp]),
    "dihost_model_set_p2p_comm": (i32, [vp, vp]),
    "dihost_weight_file_index": (i32, [C.c_char_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "dihost_weights_load_file": (i32, [vp, C.c_char_p, C.POINTER(i32)]),
    "dihost_weight_file_slice": (i32, [C.c_char_p, C.c_char_p, i32, i32, vp, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_int64), C.POINTER(i32)]),
    "dihost_get_weight": (i32, [vp, C.c_char_p, C.POINTER(i32), C.POINTER(i32), C.POINTER(C.c_int64), C.POINTER(vp)]),
    "dihost_set_tensor": (i32, [vp, C.c_char_p, i32, i32, C.POINTER(C.c_int64), vp]),
    "dihost_set_weight": (i32, [vp, C.c_char_p, i32, i32, C.POINTER(C.c_int64), vp]),
    "dihost_get_tensor": (i32, [vp, C.c_char_p, C.POINTER(i32), C.POINTER(i32), C.POINTER(C.c_int64), C.POINTER(vp)]),
    "dihost_op_create": (i32, [vp, C.POINTER(i32), C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p]),
    "dihost_set_runtime": (i32, [vp, i32, i32, C.POINTER(i32), i32, i32, C.POINTER(vp), C.POINTER(vp)]),
    "dihost_set_prefix_len": (i32, [vp, i32, i32]),
    "dihost_op_reshape": (i32, [vp, i32]),
    "dihost_op_alloc": (i32, [vp, i32]),
    "dihost_op_forward": (i32, [vp, i32]),
    "dihost_ops_alloc_concurrent": (i32, [vp, C.POINTER(i32), i32]),
    "dihost_cache_seq_len": (C.c_long, [vp, i32, i32]),
    "dihost_graph_add_op": (i32, [vp, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p]),
    "dihost_graph_add_serialized": (i32, [vp, C.c_char_p, C.c_size_t, C.c_char_p]),
    "dihost_graph_build": (i32, [vp, i32]),
    "dihost_graph_report": (C.c_char_p, [vp]),
    "dihost_graph_fuse_dry": (C.c_char_p, [vp]),
    "dihost_request_start": (i32, [vp, C.POINTER(C.c_int64), i32, i32, i32, C.c_float, C.c_float, C.c_ulonglong, i32, i32, C.POINTER(vp),
                                   C.POINTER(vp), C.POINTER(C.c_int64)]),
    "dihost_request_adopt": (i32, [vp, i32, C.c_int64, i32, i32, C.POINTER(vp), C.POINTER(vp)]),
    "dihost_request_stop": (i32, [vp, i32]),
    "dihost_decode_steps": (i32, [vp, i32, i32]),
    "dihost_sync_ids": (i32, [vp, C.POINTER(C.c_int64), i32]),
    "dihost_request_attach": (i32, [vp, i32, C.POINTER(C.c_int64), i32, i32, i32, i32, C.POINTER(C.c_int64), i32, i32, i32]),
    "dihost_request_put_token": (i32, [vp, i32, i32, C.c_int64]),
    "dihost_request_set_step": (i32, [vp, i32, i32, i32]),
    "dihost_set_phase": (i32, [vp, i32]),
    "dihost_request_poll": (i32, [vp, i32, C.POINTER(C.c_int64), i32, C.POINTER(i32), C.POINTER(i32)]),
    "dihost_next_request_generation": (i32, [vp, C.c_float, C.c_float, C.c_float, i32, i32, i32, i32, i32, i32]),
    "dihost_request_generation": (i32, [vp, i32, C.c_float, C.c_float, C.c_float, i32, i32, i32, i32, i32, i32, i32]),
    "dihost_request_logprobs": (i32, [vp, i32, i32, i32, i32, i32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(i32)]),
    "dihost_running_batch": (i32, [vp]),
    "dihost_requests_rewind": (i32, [vp, i32]),
    "dihost_last_error": (C.c_char_p, []),
    "dihost_registered_ops": (C.c_char_p, []),
}
_lib = None


def header_symbols():
    text = open(os.path.join(REPO_ROOT, "include", "dashinfer_hip_host.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dihost_[a-z0-9_]+)\s*\(", text)))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(OPS_LIB_PATH):
            raise RuntimeError(f"{OPS_LIB_PATH} is missing: build it with `make -C dash-infer_amd/host` (or __graft_entry__.build())")
        _devlib()  # libdashinfer_hip.so (and torch's HIP runtime) first
        l = C.CDLL(OPS_LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


class HostError(RuntimeError):
    def __init__(self, code, what):
        dev = _devlib().dihip_last_error()
        super().__init__(f"{what}: AsStatus {code}: {lib().dihost_last_error().decode()} (device library: {dev.decode() if dev else ''})")
        self.code = code


def _ck(code, what):
    if code != 0:
        raise HostError(code, what)


class Model:
    """HIPContext + TensorMap + RuntimeContext, driven the way AsModel drives its operators."""

    def __init__(self, stream, n_heads, n_groups, head_size, span, cache_mode=0, max_batch=1, max_len=0, rank=0, nranks=1, comm=None):
        self.h = vp()
        _ck(lib().dihost_model_create(C.byref(self.h), stream, n_heads, n_groups, head_size, span, cache_mode, max_batch, max_len,
                                      rank, nranks, comm), "dihost_model_create")
        self._keep = []

    def load_weight_file(self, path):
        """every record of a serialized weight file (.asparam) as a weight of the model, in memory the model owns -> records loaded"""
        n = i32(0)
        _ck(lib().dihost_weights_load_file(self.h, str(path).encode(), C.byref(n)), "weights_load_file")
        return int(n.value)

    def get_weight(self, name):
        """-> (dtype code, shape, device pointer)"""
        dt, nd, ptr = i32(), i32(), vp()
        shp = (C.c_int64 * 8)()
        _ck(lib().dihost_get_weight(self.h, name.encode(), C.byref(dt), C.byref(nd), shp, C.byref(ptr)), "get_weight")
        return dt.value, [int(shp[i]) for i in range(nd.value)], ptr.value

    def set_p2p_comm(self, handle):
        """the rank's one-shot peer-to-peer communicator (capi dihip_p2p_ar_create) for the AllReduce operator"""
        _ck(lib().dihost_model_set_p2p_comm(self.h, handle), "set_p2p_comm")

    def close(self):
        if self.h:
            lib().dihost_model_destroy(self.h)
            self.h = None

    def _put(self, fn, name, t, dtype):
        shape = (C.c_int64 * t.dim())(*t.shape)
        self._keep.append(t)
        _ck(fn(self.h, name.encode(), DT[dtype], t.dim(), shape, vp(t.data_ptr())), "set " + name)

    def set_tensor(self, name, t, dtype):
        self._put(lib().dihost_set_tensor, name, t, dtype)

    def set_weight(self, name, t, dtype):
        self._put(lib().dihost_set_weight, name, t, dtype)

    def get_tensor(self, name):
        dt, nd, shp, p = i32(), i32(), (C.c_int64 * 8)(), vp()
        _ck(lib().dihost_get_tensor(self.h, name.encode(), C.byref(dt), C.byref(nd), shp, C.byref(p)), "get " + name)
        return dt.value, list(shp[: nd.value]), p.value

    def create_op(self, op_type, op_name, inputs, outputs, weights=(), attrs=""):
        oid = i32()
        _ck(lib().dihost_op_create(self.h, C.byref(oid), op_type.encode(), op_name.encode(), ",".join(inputs).encode(),
                                   ",".join(outputs).encode(), ",".join(weights).encode(), attrs.encode()), "create " + op_type)
        self._names = getattr(self, "_names", {})
        self._names[oid.value] = f"{op_type} {op_name}"
        return oid.value

    def set_runtime(self, is_context, steps, k_spans, v_spans):
        """k_spans / v_spans: [request][layer][span] lists of device pointers."""
        nreq = len(steps)
        nl = len(k_spans[0]) if nreq else 0
        spr = len(k_spans[0][0]) if nl else 0
        flat = lambda a: (vp * (nreq * nl * spr))(*[p for r in a for l in r for p in l])
        _ck(lib().dihost_set_runtime(self.h, int(is_context), nreq, (i32 * max(nreq, 1))(*steps), nl, spr, flat(k_spans), flat(v_spans)),
            "set_runtime")

    def set_prefix_len(self, request, prefix_len):
        _ck(lib().dihost_set_prefix_len(self.h, request, prefix_len), "set_prefix_len")

    def reshape(self, op):
        _ck(lib().dihost_op_reshape(self.h, op), f"CallReshape [{self._names.get(op, op)}]")

    def alloc(self, op):
        _ck(lib().dihost_op_alloc(self.h, op), f"CallAlloc [{self._names.get(op, op)}]")

    def forward(self, op):
        _ck(lib().dihost_op_forward(self.h, op), f"CallForward [{self._names.get(op, op)}]")

    # ---- the model runner: the reference's operator list, optionally fused, driven like AsModel's decode loop ----------------
    def graph_add_op(self, op_type, op_name, inputs, outputs, weights=(), attrs=""):
        _ck(lib().dihost_graph_add_op(self.h, op_type.encode(), op_name.encode(), ",".join(inputs).encode(), ",".join(outputs).encode(),
                                      ",".join(weights).encode(), attrs.encode()), "graph_add_op " + op_type)

    def graph_add_serialized(self, model_bytes, graphs=None):
        """The operator lists of a serialized allspark TransformerProto (what the reference's converter writes; graph_proto.py) --
        graphs: names in order, default decoder + gen_graph, the two graphs AsModel runs per step."""
        _ck(lib().dihost_graph_add_serialized(self.h, model_bytes, len(model_bytes), ",".join(graphs).encode() if graphs else None),
            "graph_add_serialized")

    @staticmethod
    def _report(text):
        out = {}
        for item in text.decode().split(";"):
            k, _, v = item.partition("=")
            out[k] = v
        out["fused"] = out.get("fused") == "1"
        out["device_resident"] = out.get("device_resident") == "1"
        out["layers"] = int(out.get("layers", 0))
        out["types"] = [t for t in out.get("types", "").split(",") if t]
        return out

    def graph_build(self, fuse=True):
        _ck(lib().dihost_graph_build(self.h, int(fuse)), "graph_build")
        return self._report(lib().dihost_graph_report(self.h))

    def graph_fuse_dry(self):
        return self._report(lib().dihost_graph_fuse_dry(self.h))

    @staticmethod
    def _spans(k_spans, v_spans):
        nl = len(k_spans)
        spr = len(k_spans[0]) if nl else 0
        flat = lambda a: (vp * (nl * spr))(*[p for l in a for p in l])
        return nl, spr, flat(k_spans), flat(v_spans)

    def request_start(self, prompt, k_spans, v_spans, prefix_len=0, top_k=1, top_p=1.0, temperature=1.0, seed=0):
        """k_spans / v_spans: [layer][span] device pointers of this request's cache.  -> the first generated id."""
        nl, spr, ks, vs = self._spans(k_spans, v_spans)
        ids = (C.c_int64 * len(prompt))(*[int(t) for t in prompt])
        first = C.c_int64()
        _ck(lib().dihost_request_start(self.h, ids, len(prompt), prefix_len, top_k, top_p, temperature, seed, nl, spr, ks, vs, C.byref(first)),
            "request_start")
        return first.value

    def request_adopt(self, cached_len, next_id, k_spans, v_spans):
        nl, spr, ks, vs = self._spans(k_spans, v_spans)
        _ck(lib().dihost_request_adopt(self.h, cached_len, int(next_id), nl, spr, ks, vs), "request_adopt")

    def request_stop(self, index):
        _ck(lib().dihost_request_stop(self.h, index), "request_stop")

    def decode_steps(self, n=1, graph=False):
        _ck(lib().dihost_decode_steps(self.h, n, int(graph)), "decode_steps")

    def sync_ids(self):
        cap = max(1, lib().dihost_running_batch(self.h))
        buf = (C.c_int64 * cap)()
        n = lib().dihost_sync_ids(self.h, buf, cap)
        if n < 0:
            raise HostError(-n, "sync_ids")
        return [int(buf[i]) for i in range(n)]

    def requests_rewind(self, cached_len):
        _ck(lib().dihost_requests_rewind(self.h, int(cached_len)), "requests_rewind")

    # ---- the Request slice behind PreProcessId / UpdateId / PostProcessId -------------------------------------------------
    def request_attach(self, index, ids, max_length=0, early_stopping=True, eos=-1, stop_words=(), in_length_bias=0):
        arr = (C.c_int64 * len(ids))(*[int(t) for t in ids])
        wl = len(stop_words[0]) if stop_words else 0
        assert all(len(w) == wl for w in stop_words)
        flat = [int(t) for w in stop_words for t in w]
        sw = (C.c_int64 * max(1, len(flat)))(*flat)
        _ck(lib().dihost_request_attach(self.h, index, arr, len(ids), max_length, int(early_stopping), eos, sw, len(stop_words), wl,
                                        in_length_bias), "request_attach")

    def request_put_token(self, index, position, token):
        _ck(lib().dihost_request_put_token(self.h, index, position, int(token)), "request_put_token")

    def request_set_step(self, index, step, in_length_bias=0):
        _ck(lib().dihost_request_set_step(self.h, index, step, in_length_bias), "request_set_step")

    def request_poll(self, index, capacity=64):
        buf, fin, ni = (C.c_int64 * capacity)(), i32(), i32()
        n = lib().dihost_request_poll(self.h, index, buf, capacity, C.byref(fin), C.byref(ni))
        if n < 0:
            raise HostError(-n, "request_poll")
        return [int(buf[i]) for i in range(n)], bool(fin.value), ni.value

    # ---- GenerateOp's logits processors and log-probability outputs ---------------------------------------------------------
    @staticmethod
    def _gen(g):
        return (float(g.get("repetition_penalty", 1.0)), float(g.get("frequency_penalty", 0.0)), float(g.get("presence_penalty", 0.0)),
                int(g.get("no_repeat_ngram_size", 0)), int(g.get("min_length", 0)), int(g.get("eos_token_id", -1)),
                int(bool(g.get("suppress_repetition_in_generation", False))), int(bool(g.get("logprobs", False))), int(g.get("top_logprobs", 0)))

    def next_request_generation(self, **g):
        """GenerateConfig's processor / logprobs fields for the NEXT request_start (model-runner path)."""
        _ck(lib().dihost_next_request_generation(self.h, *self._gen(g)), "next_request_generation")

    def request_generation(self, index, input_len, **g):
        """... for request `index` of the runtime context (operator-by-operator path)."""
        _ck(lib().dihost_request_generation(self.h, index, *self._gen(g), int(input_len)), "request_generation")

    def request_logprobs(self, index, first, count, top_n, runner=True):
        """-> (token_logprob [n], top_value [n, top_n], top_index [n, top_n]) as lists."""
        tok = (C.c_float * max(count, 1))()
        val = (C.c_float * max(count * top_n, 1))()
        idx = (i32 * max(count * top_n, 1))()
        n = lib().dihost_request_logprobs(self.h, index, int(runner), first, count, top_n, tok, val, idx)
        if n < 0:
            raise HostError(-n, "request_logprobs")
        return ([tok[i] for i in range(n)], [[val[i * top_n + k] for k in range(top_n)] for i in range(n)],
                [[idx[i * top_n + k] for k in range(top_n)] for i in range(n)])

    def set_phase(self, is_context):
        _ck(lib().dihost_set_phase(self.h, int(is_context)), "set_phase")


def weight_file_index(path):
    """The records of a serialized weight file (.asparam) -> [(name, dtype code, shape, split_mode, offset, nbytes)]; needs no GPU."""
    need = C.c_size_t(0)
    _ck(lib().dihost_weight_file_index(str(path).encode(), None, 0, C.byref(need)), "weight_file_index")
    buf = C.create_string_buffer(need.value)
    _ck(lib().dihost_weight_file_index(str(path).encode(), buf, need.value, None), "weight_file_index")
    out = []
    for line in buf.value.decode().splitlines():
        name, dt, shape, split, off, nb = line.rsplit("|", 5)
        out.append((name, int(dt), [int(d) for d in shape.split(",") if d], int(split), int(off), int(nb)))
    return out


def weight_file_slice(path, name, rank, nranks):
    """One record's share for (rank, nranks) -- the bytes dihost_weights_load_file uploads on that rank (the reference's WeightSplitter
    rules, host/weight_file.h SliceForRank) -> (bytes, shape); needs no GPU."""
    nb, nd, shp = C.c_size_t(0), i32(0), (C.c_int64 * 8)()
    _ck(lib().dihost_weight_file_slice(str(path).encode(), name.encode(), rank, nranks, None, 0, C.byref(nb), shp, C.byref(nd)), "weight_file_slice")
    buf = C.create_string_buffer(max(1, nb.value))
    _ck(lib().dihost_weight_file_slice(str(path).encode(), name.encode(), rank, nranks, buf, nb.value, C.byref(nb), shp, C.byref(nd)), "weight_file_slice")
    return buf.raw[: nb.value], [int(shp[i]) for i in range(nd.value)]
