// weight_file.h -- reader of the reference's serialized weights (".asparam", the "allsparkz" container of the converter:
// python/pyhie/allspark/model/model_base.py -> csrc/utility/allsparkz_util.cpp:264-339).  Layout, as that writer produces it:
//
//   record*   "AS" | u16 0x0001 | u16 name length | name | header | data
//             header = one text line  {'descr': '<f4', 'fortran_order': False, 'shape': (K, N),'group_list': (),'sparse_type': 0,'nnz': 0,'split_type': 1,}\n
//                      (create_allsparky_header, :107-146: byte order, type letter, word size; shape; SplitMode of the TP splitter)
//             data   = prod(shape) * word size bytes, little endian, row-major           (dense records)
//                      sparse_type 1 (CSC, allsparkz_util.cpp:162-203): col_offset int32 [cols + 1] | row_idx int32 [nnz] | values [nnz]
//                      sparse_type 2 (ELL, :205-254):                   row_idx uint16 [nnz] | values [nnz], nnz = cols * max_c, laid out in
//                                    blocks of VECT = 16 / word entries per column (sparse_util.cpp:91-131)
//                      both for 2-D f32 / f16 matrices; DENSIFIED on the way in (ReadRecordDense): this backend has no sparse GEMM, the
//                      operators see the dense matrix the converter started from
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
#include <cstring>
#include <string>
#include <vector>

#include "as_types.h"

namespace allspark {

struct WeightRecord {
  std::string name;
  DataType dtype = DATATYPE_UNDEFINED;
  std::vector<int64_t> shape;
  int split_mode = 0;    // allspark.proto SplitMode of the tensor-parallel splitter (NOSPLIT 0, VSPLIT 1, HSPLIT 2, ...)
  std::vector<int64_t> group_list;  // GROUP_VSPLIT: widths of the column groups (qkv: [n H, g H, g H]; halved by the converter for packed int4)
  int sparse_type = 0;   // 0 dense, 1 CSC, 2 ELL
  long long nnz = 0;     // stored entries of a sparse record (padding included)
  long long offset = 0;  // of the data in the file
  long long nbytes = 0;  // of the DENSE tensor
  long long stored_bytes = 0;  // of the record's data in the file (== nbytes for a dense record)
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
  std::fseek(fp, 0, SEEK_END);
  const long long file_size = std::ftell(fp);
  std::fseek(fp, 0, SEEK_SET);
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
    std::string descr, shape, sparse, split, groups;
    if (!field(h, "descr", &descr) || descr.size() < 4 || !field(h, "shape", &shape)) return fail("unreadable header of " + r.name + ": " + h);
    if (descr[1] != '<' && descr[1] != '|') return fail(r.name + ": big-endian data");
    const int word = std::atoi(descr.substr(3).c_str());
    if (!dtype_of(descr[2], word, &r.dtype)) return fail(r.name + ": unsupported element type " + descr);
    auto ints = [](const std::string& t, std::vector<int64_t>* v) {  // the non-negative integers of "(a, b, c)"
      for (size_t i = 0; i < t.size();) {
        if (t[i] >= '0' && t[i] <= '9') {
          char* end = nullptr;
          v->push_back(std::strtoll(t.c_str() + i, &end, 10));
          i = (size_t)(end - t.c_str());
        } else {
          ++i;
        }
      }
    };
    ints(shape, &r.shape);
    if (r.shape.empty()) return fail(r.name + ": no shape");
    if (field(h, "sparse_type", &sparse)) r.sparse_type = std::atoi(sparse.c_str());
    long long count = 1;
    // (a malformed shape must not wrap the byte count.)  A dense record cannot be larger than the file; a sparse one is stored
    // compressed, so its dense size is bounded by a fixed 64 GiB instead
    const long long dense_limit = r.sparse_type == 0 ? file_size : (1ll << 36);
    for (int64_t d : r.shape) {
      if (d < 0 || (d > 0 && count > (long long)(dense_limit / (d * (long long)word)) + 1)) return fail(r.name + ": shape larger than the file");
      count *= d;
    }
    if (field(h, "split_type", &split)) r.split_mode = std::atoi(split.c_str());
    if (field(h, "group_list", &groups)) ints(groups, &r.group_list);
    r.nbytes = count * word;
    r.stored_bytes = r.nbytes;
    if (r.sparse_type != 0) {
      std::string nnz;
      if (r.sparse_type != 1 && r.sparse_type != 2) return fail(r.name + ": unknown sparse_type " + std::to_string(r.sparse_type));
      if (r.shape.size() != 2 || (r.dtype != FLOAT32 && r.dtype != FLOAT16)) return fail(r.name + ": a sparse record that is not a 2-D f32 / f16 matrix");
      if (!field(h, "nnz", &nnz)) return fail(r.name + ": sparse record without nnz");
      r.nnz = std::atoll(nnz.c_str());
      if (r.nnz < 0 || r.nnz > file_size) return fail(r.name + ": nnz larger than the file");
      if (r.sparse_type == 2 && (r.shape[1] == 0 || r.nnz % r.shape[1] != 0 || (r.nnz / r.shape[1]) % (16 / word) != 0 || r.shape[0] > 65536))
        return fail(r.name + ": malformed ELL record (nnz must be cols x a multiple of 16 / word size, at most 65536 rows)");
      r.stored_bytes = r.sparse_type == 1 ? (r.shape[1] + 1) * 4 + r.nnz * (4 + word) : r.nnz * (2 + word);
    }
    r.offset = std::ftell(fp);
    if (r.offset < 0 || r.offset + r.stored_bytes > file_size) return fail("truncated data of " + r.name);  // EVERY record against the file's size
    if (std::fseek(fp, (long)(r.offset + r.stored_bytes), SEEK_SET) != 0) return fail("truncated data of " + r.name);
    out->push_back(std::move(r));
  }
  std::fclose(fp);
  return true;
}

// reads record `r` of an open file as the DENSE row-major tensor (r.nbytes bytes): dense records as stored; CSC / ELL records scattered
// back into the matrix the reference's writer compressed (dense_to_csc_padding / dense_to_ell_padding, sparse_util.cpp:23-131).  The
// writers pad columns with ZERO-valued entries whose row index repeats the last real one (ELL: whatever was in the buffer), so entries
// ADD into a zeroed matrix and an entry outside it is an error only when its value is not zero.
inline bool ReadRecordDense(FILE* fp, const WeightRecord& r, std::vector<char>* dense, std::string* err) {
  auto fail = [&](const std::string& what) {
    *err = r.name + ": " + what;
    return false;
  };
  std::vector<char> stored((size_t)r.stored_bytes);
  if (std::fseek(fp, (long)r.offset, SEEK_SET) != 0 || (!stored.empty() && std::fread(stored.data(), 1, stored.size(), fp) != stored.size()))
    return fail("cannot read the record");
  if (r.sparse_type == 0) {
    dense->swap(stored);
    return true;
  }
  const int64_t rows = r.shape[0], cols = r.shape[1];
  const size_t word = (size_t)SizeofType(r.dtype);
  dense->assign((size_t)r.nbytes, 0);
  auto put = [&](int64_t row, int64_t col, const char* v) {  // dense[row, col] = v unless v is +0 / -0 (padding)
    bool zero = true;
    for (size_t b = 0; b < word; ++b) zero = zero && (v[b] == 0 || (b == word - 1 && (unsigned char)v[b] == 0x80));
    if (zero) return true;
    if (row < 0 || row >= rows || col < 0 || col >= cols) return false;
    std::memcpy(dense->data() + ((size_t)row * cols + col) * word, v, word);
    return true;
  };
  if (r.sparse_type == 1) {  // CSC: col_offset [cols + 1] | row_idx [nnz] | values [nnz]
    const int32_t* off = reinterpret_cast<const int32_t*>(stored.data());
    const int32_t* ridx = off + cols + 1;
    const char* val = reinterpret_cast<const char*>(ridx + r.nnz);
    if (off[0] != 0 || off[cols] != r.nnz) return fail("CSC column offsets do not span nnz");
    for (int64_t c = 0; c < cols; ++c) {
      if (off[c + 1] < off[c]) return fail("CSC column offsets decrease");
      for (int64_t e = off[c]; e < off[c + 1]; ++e)
        if (!put(ridx[e], c, val + (size_t)e * word)) return fail("CSC entry outside the matrix");
    }
    return true;
  }
  // ELL: blocks of VECT entries per column: for b in [0, max_c / VECT): for col: VECT entries (sparse_util.cpp:119-130)
  const int64_t vect = 16 / (int64_t)word, max_c = cols ? r.nnz / cols : 0;
  const uint16_t* ridx = reinterpret_cast<const uint16_t*>(stored.data());
  const char* val = reinterpret_cast<const char*>(ridx + r.nnz);
  int64_t pos = 0;
  for (int64_t b = 0; b < max_c / vect; ++b)
    for (int64_t c = 0; c < cols; ++c)
      for (int64_t k = 0; k < vect; ++k, ++pos)
        if (!put(ridx[pos], c, val + (size_t)pos * word)) return fail("ELL entry outside the matrix");
  return true;
}

// ---- the tensor-parallel split at load time (WeightManager -> WeightSplitter, csrc/runtime/weight/weight_splitter.cpp) ------------------
// `whole` = the record's bytes as stored; -> this rank's share (`out`, `out_shape`), by the record's SplitMode:
//   NOSPLIT                       the whole tensor on every rank
//   VSPLIT (:60-127)              2-D [K, N]: columns [rank N/R, (rank + 1) N/R);  1-D [N]: the same range
//   HSPLIT (:369-438)             2-D: rows [rank K/R, ...);  1-D (a bias added after the row-parallel sum): whole on rank 0, ZEROS elsewhere
//   GROUP_VSPLIT (:611-721)       per group of `group_list` (qkv: q | k | v): the rank's 1/R of every group, concatenated (2-D and 1-D)
//   BATCH_VSPLIT (:128-232)       3-D [E, K, N]: columns of every matrix;  2-D [E, N]: columns of every row (per-expert scales)
//   BATCH_HSPLIT (:439-520)       3-D [E, K, N]: rows of every matrix
//   QKVSPLIT / KVSPLIT (:521-610) three / two equal column groups, the rank's share of each;  MQA_VSPLIT (:722-852) [q | k | v]: q split, k / v whole
//   EPSPLIT (:853-919)            3-D [E, K, N]: the rank's experts
// Everything must divide by the rank count, as IsSplittable demands; BATCH_KVSPLIT and HSPLIT_QUANTIZE (no splitter in the factory,
// :921-960) are refused for nranks > 1.
inline bool SliceForRank(const WeightRecord& r, const char* whole, int rank, int nranks, std::vector<char>* out, std::vector<int64_t>* out_shape,
                         std::string* err) {
  const size_t word = r.shape.empty() ? 0 : (size_t)SizeofType(r.dtype);
  auto fail = [&](const std::string& what) {
    *err = r.name + ": " + what;
    return false;
  };
  *out_shape = r.shape;
  if (nranks <= 1 || r.split_mode == 0) {
    out->assign(whole, whole + r.nbytes);
    return true;
  }
  if (rank < 0 || rank >= nranks) return fail("rank outside the group");
  const int nd = (int)r.shape.size();
  // the tensor as [outer][rows][cols]: 1-D = one row; 2-D = [rows, cols]; 3-D = [outer, rows, cols]
  const int64_t cols = r.shape[nd - 1], rows = nd >= 2 ? r.shape[nd - 2] : 1, outer = nd >= 3 ? r.shape[0] : 1;
  if (nd > 3) return fail("split of a rank-" + std::to_string(nd) + " tensor");
  auto take_cols = [&](const std::vector<std::pair<int64_t, int64_t>>& ranges) {  // [(first column, count)] of every row, concatenated
    int64_t w = 0;
    for (auto& g : ranges) w += g.second;
    out->resize((size_t)(outer * rows * w) * word);
    char* d = out->data();
    for (int64_t o = 0; o < outer * rows; ++o)
      for (auto& g : ranges) {
        std::memcpy(d, whole + ((size_t)o * cols + g.first) * word, (size_t)g.second * word);
        d += (size_t)g.second * word;
      }
    (*out_shape)[nd - 1] = w;
  };
  auto take_rows = [&](int64_t first, int64_t count) {
    out->resize((size_t)(outer * count * cols) * word);
    for (int64_t o = 0; o < outer; ++o)
      std::memcpy(out->data() + (size_t)(o * count * cols) * word, whole + ((size_t)(o * rows + first) * cols) * word, (size_t)(count * cols) * word);
    (*out_shape)[nd - 2] = count;
  };
  switch (r.split_mode) {
    case 1:  // VSPLIT
      if (nd > 2) return fail("VSPLIT of a rank-3 tensor (BATCH_VSPLIT is the batched form)");
      if (cols % nranks) return fail("VSPLIT: " + std::to_string(cols) + " columns do not divide by " + std::to_string(nranks) + " ranks");
      take_cols({{rank * (cols / nranks), cols / nranks}});
      return true;
    case 2:  // HSPLIT
      if (nd > 2) return fail("HSPLIT of a rank-3 tensor (BATCH_HSPLIT is the batched form)");
      if (nd == 1) {  // the bias of a row-parallel layer: added once, on rank 0
        if (rank == 0) out->assign(whole, whole + r.nbytes);
        else out->assign((size_t)r.nbytes, 0);
        return true;
      }
      if (rows % nranks) return fail("HSPLIT: " + std::to_string(rows) + " rows do not divide by " + std::to_string(nranks) + " ranks");
      take_rows(rank * (rows / nranks), rows / nranks);
      return true;
    case 6: {  // GROUP_VSPLIT
      if (nd > 2) return fail("GROUP_VSPLIT of a rank-3 tensor");
      if (r.group_list.empty()) return fail("GROUP_VSPLIT without a group_list");
      std::vector<std::pair<int64_t, int64_t>> ranges;
      int64_t at = 0;
      for (int64_t g : r.group_list) {
        if (g % nranks) return fail("GROUP_VSPLIT: a group of " + std::to_string(g) + " columns does not divide by " + std::to_string(nranks) + " ranks");
        ranges.push_back({at + rank * (g / nranks), g / nranks});
        at += g;
      }
      if (at != cols) return fail("GROUP_VSPLIT: the group_list sums to " + std::to_string(at) + ", the tensor has " + std::to_string(cols) + " columns");
      take_cols(ranges);
      return true;
    }
    case 8:  // BATCH_VSPLIT: every matrix (3-D) / every row (2-D) by columns
      if (nd < 2) return fail("BATCH_VSPLIT of a vector");
      if (cols % nranks) return fail("BATCH_VSPLIT: " + std::to_string(cols) + " columns do not divide by " + std::to_string(nranks) + " ranks");
      take_cols({{rank * (cols / nranks), cols / nranks}});
      return true;
    case 3:    // QKVSPLIT: three equal column groups (WeightSplitterVSplitBatchGEMM<3>, :521-610)
    case 4: {  // KVSPLIT: two
      const int cnt = r.split_mode == 3 ? 3 : 2;
      if (nd > 2) return fail("QKVSPLIT / KVSPLIT of a rank-3 tensor");
      if (cols % (cnt * nranks)) return fail(std::to_string(cols) + " columns do not divide by " + std::to_string(cnt) + " x " + std::to_string(nranks));
      std::vector<std::pair<int64_t, int64_t>> ranges;
      const int64_t g = cols / cnt;
      for (int i = 0; i < cnt; ++i) ranges.push_back({i * g + rank * (g / nranks), g / nranks});
      take_cols(ranges);
      return true;
    }
    case 7: {  // MQA_VSPLIT (:722-852): [q | k | v] with the q columns split and the one K / V head on every rank
      if (nd > 2) return fail("MQA_VSPLIT of a rank-3 tensor");
      if (r.group_list.size() != 3) return fail("MQA_VSPLIT needs a group_list of three");
      if (r.group_list[0] % nranks) return fail("MQA_VSPLIT: " + std::to_string(r.group_list[0]) + " query columns do not divide by " + std::to_string(nranks) + " ranks");
      if (r.group_list[0] + r.group_list[1] + r.group_list[2] != cols) return fail("MQA_VSPLIT: the group_list does not sum to the column count");
      const int64_t q = r.group_list[0] / nranks;
      take_cols({{rank * q, q}, {r.group_list[0], r.group_list[1]}, {r.group_list[0] + r.group_list[1], r.group_list[2]}});
      return true;
    }
    case 11: {  // EPSPLIT (:853-919): the experts [rank E/R, ...) of a [E, K, N] stack, whole matrices
      if (nd != 3) return fail("EPSPLIT of a tensor that is not [experts, K, N]");
      if (outer % nranks) return fail("EPSPLIT: " + std::to_string(outer) + " experts do not divide by " + std::to_string(nranks) + " ranks");
      const size_t per = (size_t)(outer / nranks) * rows * cols * word;
      out->assign(whole + (size_t)rank * per, whole + (size_t)(rank + 1) * per);
      (*out_shape)[0] = outer / nranks;
      return true;
    }
    case 9:  // BATCH_HSPLIT (the reference takes rank-3 tensors only; the converter writes the parameters of such experts NOSPLIT)
      if (nd != 3) return fail("BATCH_HSPLIT of a tensor that is not [experts, K, N]");
      if (rows % nranks) return fail("BATCH_HSPLIT: " + std::to_string(rows) + " rows do not divide by " + std::to_string(nranks) + " ranks");
      take_rows(rank * (rows / nranks), rows / nranks);
      return true;
    default:
      return fail("SplitMode " + std::to_string(r.split_mode) + " is not served under tensor parallelism by this backend");
  }
}

}  // namespace allspark
