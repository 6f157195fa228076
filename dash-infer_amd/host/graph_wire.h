// graph_wire.h -- reads the operator lists out of a SERIALIZED allspark `TransformerProto` (csrc/proto/allspark.proto:133-159), the
// bytes the reference's converter writes (python/pyhie/allspark/model/*.py -> model.SerializeToString()) and AsModel parses with
// the generated protobuf classes before it creates the operators (csrc/core/model/model.cpp:265-287).  The HIP operator layer's
// stand-in OperatorProto (as_types.h) keeps tensor NAMES and raw attribute bytes, which is all InitV2 reads -- so a
// 90-line wire-format reader replaces the protobuf dependency (absent in this build):
//   TransformerProto: 6 = map<string, GraphProto> graphs, 7 = repeated string graph_names
//   GraphProto:       3 = repeated OperatorProto ops
//   OperatorProto:    1 = op_type, 2 = op_name, 3 = map<string, bytes> attr, 4 / 5 / 6 = repeated TensorProto inputs / outputs / weights
//   TensorProto:      1 = name (2 = data: ignored, weights are bound by name from the weight map)
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <string_view>
#include <vector>

#include "as_types.h"

namespace allspark {
namespace wire {

struct Reader {
  const uint8_t* p;
  const uint8_t* e;
  explicit Reader(std::string_view v) : p(reinterpret_cast<const uint8_t*>(v.data())), e(p + v.size()) {}
  bool done() const { return p >= e; }
  bool varint(uint64_t& v) {
    v = 0;
    for (int shift = 0; p < e && shift < 64; shift += 7) {
      const uint8_t b = *p++;
      v |= (uint64_t)(b & 0x7F) << shift;
      if (!(b & 0x80)) return true;
    }
    return false;
  }
  bool tag(int& field, int& wt) {
    uint64_t t;
    if (!varint(t)) return false;
    field = (int)(t >> 3);
    wt = (int)(t & 7);
    return true;
  }
  bool bytes(std::string_view& out) {
    uint64_t n;
    if (!varint(n) || n > (uint64_t)(e - p)) return false;
    out = std::string_view(reinterpret_cast<const char*>(p), (size_t)n);
    p += n;
    return true;
  }
  bool skip(int wt) {
    uint64_t v;
    std::string_view s;
    switch (wt) {
      case 0: return varint(v);
      case 1: if (e - p < 8) return false; p += 8; return true;
      case 2: return bytes(s);
      case 5: if (e - p < 4) return false; p += 4; return true;
      default: return false;
    }
  }
};

// fields `a` and `b` (both length-delimited) of a map entry / TensorProto
inline bool two_fields(std::string_view v, int fa, std::string_view* a, int fb, std::string_view* b) {
  Reader r(v);
  int f, wt;
  while (!r.done()) {
    if (!r.tag(f, wt)) return false;
    std::string_view s;
    if (wt == 2 && (f == fa || f == fb)) {
      if (!r.bytes(s)) return false;
      if (f == fa && a) *a = s;
      if (f == fb && b) *b = s;
    } else if (!r.skip(wt)) {
      return false;
    }
  }
  return true;
}

inline bool parse_operator(std::string_view v, OperatorProto& op) {
  Reader r(v);
  int f, wt;
  while (!r.done()) {
    if (!r.tag(f, wt)) return false;
    if (wt != 2) {
      if (!r.skip(wt)) return false;
      continue;
    }
    std::string_view s;
    if (!r.bytes(s)) return false;
    if (f == 1) op.op_type = std::string(s);
    else if (f == 2) op.op_name = std::string(s);
    else if (f == 3) {
      std::string_view k, val;
      if (!two_fields(s, 1, &k, 2, &val)) return false;
      op.attr[std::string(k)] = std::string(val);
    } else if (f >= 4 && f <= 6) {
      std::string_view name;
      if (!two_fields(s, 1, &name, -1, nullptr)) return false;
      (f == 4 ? op.inputs : f == 5 ? op.outputs : op.weights).push_back(std::string(name));
    }
  }
  return true;
}

inline bool parse_graph(std::string_view v, std::vector<OperatorProto>& ops) {
  Reader r(v);
  int f, wt;
  while (!r.done()) {
    if (!r.tag(f, wt)) return false;
    if (wt == 2 && f == 3) {
      std::string_view s;
      OperatorProto op;
      if (!r.bytes(s) || !parse_operator(s, op)) return false;
      ops.push_back(std::move(op));
    } else if (!r.skip(wt)) {
      return false;
    }
  }
  return true;
}

// graphs by name + graph_names in file order
inline bool parse_transformer(std::string_view v, std::map<std::string, std::vector<OperatorProto>>& graphs, std::vector<std::string>& names) {
  Reader r(v);
  int f, wt;
  while (!r.done()) {
    if (!r.tag(f, wt)) return false;
    std::string_view s;
    if (wt == 2 && f == 6) {
      std::string_view k, g;
      if (!r.bytes(s) || !two_fields(s, 1, &k, 2, &g)) return false;
      if (!parse_graph(g, graphs[std::string(k)])) return false;
    } else if (wt == 2 && f == 7) {
      if (!r.bytes(s)) return false;
      names.push_back(std::string(s));
    } else if (!r.skip(wt)) {
      return false;
    }
  }
  return true;
}

}  // namespace wire
}  // namespace allspark
