// oracle/ref_asparam_shim/allspark.h -- TEST INFRASTRUCTURE ONLY.  Stand-in for csrc/interface/allspark.h when the reference's weight-file
// WRITER (csrc/utility/allsparkz_util.cpp, compiled from where it lies: oracle/Makefile target refasparam) is built alone: the one type
// that file needs from the engine's public header -- the field list of `struct TensorAttribute` (csrc/interface/allspark.h:309-317), an
// interface that has to match.
#pragma once
#include <string>
#include <vector>
namespace allspark {
struct TensorAttribute {
  int sparse_type = 0;
  int split_mode = 0;
  std::vector<int> shape;
  std::vector<int> group_list;
  char dtype;
  int word_size;
  int nnz = 0;
};
}  // namespace allspark
