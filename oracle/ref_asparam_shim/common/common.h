// oracle/ref_asparam_shim/common/common.h -- TEST INFRASTRUCTURE ONLY.  Stand-in for csrc/common/common.h: the standard headers the
// reference's allsparkz_util.cpp / sparse_util.cpp expect to come with it.
#pragma once
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
