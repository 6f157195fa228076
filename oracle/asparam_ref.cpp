// oracle/asparam_ref.cpp -- TEST INFRASTRUCTURE ONLY (tests/ and tests/golden/make_asparam_golden.py link it; the product never does).
// C entry points around the REFERENCE'S OWN weight-file writer -- allspark::util::save_allsparky_tofile / set_global_header of
// csrc/utility/allsparkz_util.cpp:306-339, compiled from where it lies (oracle/Makefile target refasparam) -- so that a test can
// write an .asparam file exactly as the reference's converter does (python/pyhie/allspark/model/model_base.py: one record per weight,
// then the global header) and hand it to the product's reader (dash-infer_amd/host/weight_file.h).
#include <cstdint>
#include <cstring>

#include "allsparkz_util.h"

extern "C" {

// appends one dense tensor record; dtype_char / word_size as the converter writes them ('f' 4, 'f' 2, 'b' 2, 'i' 1, 'u' 1, 'i' 8, ...)
int ref_asparam_append(const char* path, const char* name, const void* data, int64_t nbytes, char dtype_char, int word_size, const int* shape,
                       int ndim, int split_mode) {
  allspark::TensorAttribute info;
  info.sparse_type = 0;
  info.split_mode = split_mode;
  info.shape.assign(shape, shape + ndim);
  info.dtype = dtype_char;
  info.word_size = word_size;
  info.nnz = 0;
  try {
    allspark::util::save_allsparky_tofile(path, name, const_cast<void*>(data), nbytes, info);
  } catch (...) {
    return 1;
  }
  return 0;
}

// the same with the tensor's group_list (GROUP_VSPLIT / MQA_VSPLIT tensors: model_base.py save_torch_to_allsparky(..., group_list))
int ref_asparam_append_groups(const char* path, const char* name, const void* data, int64_t nbytes, char dtype_char, int word_size,
                              const int* shape, int ndim, int split_mode, const int* group_list, int ngroups) {
  allspark::TensorAttribute info;
  info.sparse_type = 0;
  info.split_mode = split_mode;
  info.shape.assign(shape, shape + ndim);
  info.group_list.assign(group_list, group_list + ngroups);
  info.dtype = dtype_char;
  info.word_size = word_size;
  info.nnz = 0;
  try {
    allspark::util::save_allsparky_tofile(path, name, const_cast<void*>(data), nbytes, info);
  } catch (...) {
    return 1;
  }
  return 0;
}

// a 2-D f32 matrix through the writer's SPARSE encodings (sparse_type 1 = CSC, 2 = ELL: save_allsparky, allsparkz_util.cpp:162-254, over
// dense_to_csc_padding / dense_to_ell_padding of sparse_util.cpp); -> the nnz the writer stored, or -1
int ref_asparam_append_sparse(const char* path, const char* name, const void* data, int64_t nbytes, char dtype_char, int word_size, const int* shape,
                              int ndim, int split_mode, int sparse_type) {
  allspark::TensorAttribute info;
  info.sparse_type = sparse_type;
  info.split_mode = split_mode;
  info.shape.assign(shape, shape + ndim);
  info.dtype = dtype_char;
  info.word_size = word_size;
  info.nnz = 0;
  try {
    allspark::util::save_allsparky_tofile(path, name, const_cast<void*>(data), nbytes, info);
  } catch (...) {
    return -1;
  }
  return info.nnz;
}

int ref_asparam_finish(const char* path) {
  try {
    allspark::util::set_global_header(path);
  } catch (...) {
    return 1;
  }
  return 0;
}

}  // extern "C"
