"""The reference's serialized model graph (csrc/proto/allspark.proto) without protoc: the message descriptors are built at
import time with the installed google.protobuf runtime, field for field as the .proto declares them (line numbers cited), so that
a `TransformerProto` written by the reference's own converter (python/pyhie/allspark/model/*.py -> model.SerializeToString())
parses here, and an operator list assembled here serializes to bytes the reference's C++ (`AsModel`, csrc/core/model/model.cpp:
265-287: `for (auto& op_proto : graph.ops()) ... OpFactory ... InitV2`) would read.

Used by: tests/golden/make_graph_golden.py (runs the reference's `Qwen_v15._build_graph` over these classes and commits the bytes),
hostapi.Model.graph_add_serialized / ref_graph.to_transformer_proto (feed the C++ operator layer from serialized bytes; the C++
side has its own wire-format reader, host/graph_wire.h).  Pure Python, no GPU."""
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

_F = descriptor_pb2.FieldDescriptorProto
_T = {"double": _F.TYPE_DOUBLE, "float": _F.TYPE_FLOAT, "int64": _F.TYPE_INT64, "int32": _F.TYPE_INT32, "bool": _F.TYPE_BOOL,
      "string": _F.TYPE_STRING, "bytes": _F.TYPE_BYTES}

# allspark.proto:9-92
ENUMS = {
    "DeviceType": {"DEVICETYPE_UNDEFINED": 0, "CPU": 1, "CUDA": 2, "COMPILE_TIME_MAX_DEVICE_TYPES": 3, "CPU_PINNED": 4},
    "DataMode": {"DENSE": 0, "CSC": 1, "ELL": 2},
    "SplitMode": {"NOSPLIT": 0, "VSPLIT": 1, "HSPLIT": 2, "QKVSPLIT": 3, "KVSPLIT": 4, "HSPLIT_QUANTIZE": 5, "GROUP_VSPLIT": 6, "MQA_VSPLIT": 7,
                  "BATCH_VSPLIT": 8, "BATCH_HSPLIT": 9, "BATCH_KVSPLIT": 10, "EPSPLIT": 11},
    "DataType": {"DATATYPE_UNDEFINED": 0, "FLOAT32": 1, "FLOAT16": 2, "INT8": 3, "INT16": 4, "INT32": 5, "INT64": 6, "STRING": 7, "BOOL": 8,
                 "BFLOAT16": 9, "UINT8": 10, "FLOAT8E4M3": 11, "FLOAT8E5M2": 12, "POINTER": 20},
    "PrecisionLevel": {"HIGHEST": 0, "HIGH": 1, "MEDIUM_BF16": 2, "MEDIUM_FP16": 3},
    "BinaryType": {"BINARYTYPE_UNDEFINED": 0, "ADD": 1, "MUL": 2, "FUSED_MUL_ADD_1": 10, "GEGLU": 11, "SWIGLU": 12},
    "UnaryType": {"UNARYTYPE_UNDEFINED": 0, "TANH": 1, "GELU_ERF": 2, "GELU_TANH": 3, "RELU": 4, "SILU": 5, "SIGMOID": 6},
    "RotaryInvFreqType": {"base_rotary": 0, "chatglm_v2": 1, "chatglm_v3": 2, "yarn": 3},
}

# message -> [(name, number, type, label)]; type: scalar name | "enum:X" | "msg:X" | "map:<key scalar>,<value type>"   (allspark.proto:94-159)
MESSAGES = {
    "ConfigProto": [("dtype", 1, "enum:DataType"), ("ln_eps", 2, "float"), ("num_heads", 3, "int32"), ("with_weights", 4, "bool"),
                    ("enc_layer", 5, "int32"), ("dec_layer", 6, "int32"), ("is_generate", 7, "bool"), ("start_dec_id", 8, "int64"),
                    ("end_dec_id", 9, "int64"), ("num_beam", 10, "int32"), ("data_mode", 11, "int64"), ("activation", 12, "enum:UnaryType"),
                    ("d_model", 13, "int32"), ("enc_num_heads", 14, "int32"), ("dec_num_heads", 15, "int32"),
                    ("multi_query_group_num", 16, "int32"), ("kv_channels", 17, "int32"), ("size_per_head", 18, "int32"),
                    ("hidden_size", 19, "int32"), ("num_experts", 20, "int32"), ("num_experts_per_tok", 21, "int32"),
                    ("intermediate_size", 22, "int32")],
    "BuildVersion": [("major", 1, "int32"), ("minor", 2, "int32"), ("patch", 3, "int32"), ("git_commit", 4, "string"), ("git_tag", 5, "string")],
    "WeightHash": [("algorithm", 1, "string"), ("hash_length", 2, "int64", "repeated"), ("hash", 3, "string", "repeated")],
    "BuildMetaProto": [("version", 1, "msg:BuildVersion"), ("weight_hash", 2, "msg:WeightHash"), ("torch_build_config", 3, "map:string,string")],
    "TransformerProto": [("model_type", 1, "string"), ("model_conf", 2, "msg:ConfigProto"), ("inputs", 3, "msg:TensorProto", "repeated"),
                         ("outputs", 4, "msg:TensorProto", "repeated"), ("weights", 5, "map:string,msg:TensorProto"),
                         ("graphs", 6, "map:string,msg:GraphProto"), ("graph_names", 7, "string", "repeated"),
                         ("build_meta", 8, "msg:BuildMetaProto")],
    "GraphProto": [("inputs", 1, "msg:TensorProto", "repeated"), ("outputs", 2, "msg:TensorProto", "repeated"),
                   ("ops", 3, "msg:OperatorProto", "repeated")],
    "OperatorProto": [("op_type", 1, "string"), ("op_name", 2, "string"), ("attr", 3, "map:string,bytes"),
                      ("inputs", 4, "msg:TensorProto", "repeated"), ("outputs", 5, "msg:TensorProto", "repeated"),
                      ("weights", 6, "msg:TensorProto", "repeated")],
    "TensorProto": [("name", 1, "string"), ("data", 2, "bytes")],
}


def _set_type(f, t):
    if t.startswith("enum:"):
        f.type, f.type_name = _F.TYPE_ENUM, ".allspark." + t[5:]
    elif t.startswith("msg:"):
        f.type, f.type_name = _F.TYPE_MESSAGE, ".allspark." + t[4:]
    else:
        f.type = _T[t]


def _build():
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name, fd.package, fd.syntax = "allspark_runtime.proto", "allspark", "proto3"
    for en, vals in ENUMS.items():
        e = fd.enum_type.add()
        e.name = en
        for k, v in vals.items():
            x = e.value.add()
            x.name, x.number = k, v
    for mn, fields in MESSAGES.items():
        m = fd.message_type.add()
        m.name = mn
        for spec in fields:
            name, num, t = spec[:3]
            f = m.field.add()
            f.name, f.number = name, num
            f.label = _F.LABEL_REPEATED if (len(spec) > 3 or t.startswith("map:")) else _F.LABEL_OPTIONAL
            if t.startswith("map:"):
                kt, vt = t[4:].split(",", 1)
                entry = m.nested_type.add()
                entry.name = "".join(p.capitalize() for p in name.split("_")) + "Entry"
                entry.options.map_entry = True
                for nm, no, tt in (("key", 1, kt), ("value", 2, vt)):
                    ef = entry.field.add()
                    ef.name, ef.number, ef.label = nm, no, _F.LABEL_OPTIONAL
                    _set_type(ef, tt)
                f.type, f.type_name = _F.TYPE_MESSAGE, f".allspark.{mn}.{entry.name}"
            else:
                _set_type(f, t)
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return pool


_POOL = _build()


def message_class(name):
    return message_factory.GetMessageClass(_POOL.FindMessageTypeByName("allspark." + name))


TensorProto = message_class("TensorProto")
OperatorProto = message_class("OperatorProto")
GraphProto = message_class("GraphProto")
TransformerProto = message_class("TransformerProto")
ConfigProto = message_class("ConfigProto")
BuildVersion = message_class("BuildVersion")
WeightHash = message_class("WeightHash")
BuildMetaProto = message_class("BuildMetaProto")


def ops_of(model_bytes, graph_names=None):
    """The operator lists of a serialized TransformerProto in the order AsModel builds them (graph_names order,
    model.cpp:265-287): [(graph name, [OperatorProto ...])]."""
    m = TransformerProto()
    m.ParseFromString(model_bytes)
    names = list(graph_names) if graph_names is not None else list(m.graph_names)
    return [(n, list(m.graphs[n].ops)) for n in names]


def op_as_tuple(op):
    """OperatorProto -> (op_type, op_name, inputs, outputs, weights, {attr: raw bytes}): tensor NAMES only (weights are bound by
    name from the weight map, as WeightManager hands them to InitV2)."""
    return (op.op_type, op.op_name, [t.name for t in op.inputs], [t.name for t in op.outputs], [t.name for t in op.weights], dict(op.attr))
