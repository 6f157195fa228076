// weight_file.h -- reader of the reference's serialized weights (".asparam", the "allsparkz" container of the converter:
// python/pyhie/allspark/model/model_base.py -> csrc/utility/allsparkz_util.cpp:264-339).  Layout, as that writer produces it:
//
//   record*   "AS" | u16 0x0001 | u16 name length | name | header | data
//             header = one text line  {'descr': '<f4', 'fortran_order': False, 'shape': (K, N),'group_list': (),'sparse_type': 0,'nnz': 0,'split_type': 1,}\n
//                      (create_allsparky_header, :107-146: byte order, type letter, word size; shape; SplitMode of the TP splitter)
//             data   = prod(shape) * word size bytes, little endian, row-major           (dense records; the sparse encodings are refused)
//   end       "AS" | u16 0 | u16 0                                                        (set_global_header, :331-339)
//
// The reference's reader is WeightFileParser (csrc/runtime/weight/weight_loader.cpp:20-130: the same letter / word-size -> DataType
// table).  This header only indexes a file (no device work); host_capi.cpp uploads the records as OWNED weight tensors, so that the
// weight-only operators free them once re-laid-out (PackedLowp::Pack).  Pinned by tests/test_host_asparam.py against files written by
// the reference's own writer (oracle/asparam_ref.cpp).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "as_types.h"

namespace allspark {

struct WeightRecord {
  std::string name;
  DataType dtype = DATATYPE_UNDEFINED;
  std::vector<int64_t> shape;
  int split_mode = 0;    // allspark.proto SplitMode of the tensor-parallel splitter (NOSPLIT 0, VSPLIT 1, HSPLIT 2, ...)
  int sparse_type = 0;
  long long offset = 0;  // of the data in the file
  long long nbytes = 0;
};

namespace weight_file_detail {

inline bool field(const std::string& h, const char* key, std::string* out) {  // the text after "'key':" up to the matching ',' / ')' / '}'
  const size_t p = h.find(std::string("'") + key + "'");
  if (p == std::string::npos) return false;
  size_t q = h.find(':', p);
  if (q == std::string::npos) return false;
  ++q;
  while (q < h.size() && h[q] == ' ') ++q;
  size_t e = q;
  if (e < h.size() && h[e] == '(') {
    e = h.find(')', e);
    if (e == std::string::npos) return false;
    ++e;
  } else if (e < h.size() && h[e] == '\'') {
    e = h.find('\'', e + 1);
    if (e == std::string::npos) return false;
    ++e;
  } else {
    while (e < h.size() && h[e] != ',' && h[e] != '}') ++e;
  }
  *out = h.substr(q, e - q);
  return true;
}

inline bool dtype_of(char letter, int word, DataType* dt) {  // weight_loader.cpp:59-120
  switch (letter) {
    case 'f': *dt = word == 4 ? FLOAT32 : word == 2 ? FLOAT16 : DATATYPE_UNDEFINED; break;  // (fp8 weights: not served by this backend)
    case 'i': *dt = word == 8 ? INT64 : word == 4 ? INT32 : word == 2 ? INT16 : word == 1 ? INT8 : DATATYPE_UNDEFINED; break;
    case 'u': *dt = word == 1 ? UINT8 : DATATYPE_UNDEFINED; break;
    case 'b': *dt = word == 2 ? BFLOAT16 : word == 1 ? BOOL : DATATYPE_UNDEFINED; break;
    default: *dt = DATATYPE_UNDEFINED;
  }
  return *dt != DATATYPE_UNDEFINED;
}

}  // namespace weight_file_detail

// indexes `path`; false + *err on a malformed or unsupported file (nothing partial is returned)
inline bool IndexWeightFile(const std::string& path, std::vector<WeightRecord>* out, std::string* err) {
  using namespace weight_file_detail;
  out->clear();
  FILE* fp = std::fopen(path.c_str(), "rb");
  if (!fp) {
    *err = "cannot open " + path;
    return false;
  }
  auto fail = [&](const std::string& what) {
    *err = path + ": " + what;
    std::fclose(fp);
    out->clear();
    return false;
  };
  for (;;) {
    unsigned char hd[6];
    if (std::fread(hd, 1, 6, fp) != 6) return fail("truncated: no global header (\"AS\" 0 0) at the end");
    if (hd[0] != 'A' || hd[1] != 'S') return fail("bad record magic");
    const unsigned flag = hd[2] | (hd[3] << 8), nlen = hd[4] | (hd[5] << 8);
    if (flag == 0 && nlen == 0) break;  // global header: end of the container
    if (flag != 1) return fail("unknown record flag " + std::to_string(flag));
    WeightRecord r;
    r.name.resize(nlen);
    if (nlen && std::fread(&r.name[0], 1, nlen, fp) != nlen) return fail("truncated record name");
    std::string h;
    for (int c; (c = std::fgetc(fp)) != EOF;) {
      h.push_back((char)c);
      if (c == '\n') break;
      if (h.size() > 4096) return fail("record header of " + r.name + " does not end");
    }
    if (h.empty() || h.back() != '\n') return fail("truncated record header of " + r.name);
    std::string descr, shape, sparse, split;
    if (!field(h, "descr", &descr) || descr.size() < 4 || !field(h, "shape", &shape)) return fail("unreadable header of " + r.name + ": " + h);
    if (descr[1] != '<' && descr[1] != '|') return fail(r.name + ": big-endian data");
    const int word = std::atoi(descr.substr(3).c_str());
    if (!dtype_of(descr[2], word, &r.dtype)) return fail(r.name + ": unsupported element type " + descr);
    long long count = 1;
    for (size_t i = 0; i < shape.size();) {
      if (shape[i] >= '0' && shape[i] <= '9') {
        char* end = nullptr;
        const long long d = std::strtoll(shape.c_str() + i, &end, 10);
        r.shape.push_back(d);
        count *= d;
        i = (size_t)(end - shape.c_str());
      } else {
        ++i;
      }
    }
    if (r.shape.empty()) return fail(r.name + ": no shape");
    if (field(h, "sparse_type", &sparse)) r.sparse_type = std::atoi(sparse.c_str());
    if (field(h, "split_type", &split)) r.split_mode = std::atoi(split.c_str());
    if (r.sparse_type != 0) return fail(r.name + ": sparse encodings (CSC / ELL) are not served by this backend");
    r.nbytes = count * word;
    r.offset = std::ftell(fp);
    if (std::fseek(fp, (long)r.nbytes, SEEK_CUR) != 0) return fail("truncated data of " + r.name);
    out->push_back(std::move(r));
  }
  // (a data block cut short shows as a bad magic or a missing global header above; check the last one explicitly)
  if (!out->empty()) {
    std::fseek(fp, 0, SEEK_END);
    const long long size = std::ftell(fp);
    if (out->back().offset + out->back().nbytes + 6 > size) return fail("truncated data of " + out->back().name);
  }
  std::fclose(fp);
  return true;
}

}  // namespace allspark
