"""dash-infer_amd -- MI355X (gfx950) device backend for DashInfer's quantized decode hot path.

The product is the C-ABI shared library ``lib/libdashinfer_hip.so`` (sources under ``csrc/``,
declarations in ``include/dashinfer_hip.h``) and the C++ operator layer under ``host/`` that
mirrors ``allspark::AsOperator``.  The Python modules here are plumbing only: a ctypes loader
(`capi`), tensor-level wrappers used by tests / bench (`ops`), the tensor-parallel weight
partitioner (`tp`) and a decode-step runner for the benchmark (`decoder`).

There is no CPU fallback: importing `capi` raises if the HIP library has not been built, and no
module in this package imports ``oracle``.
"""
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(PKG_DIR)
# DIHIP_LIB_DIR: another build of the same library (lib/trace: the GEMV kernels with their wall-clock stamps compiled in)
LIB_PATH = os.path.join(os.environ.get("DIHIP_LIB_DIR") or os.path.join(PKG_DIR, "lib"), "libdashinfer_hip.so")
OPS_LIB_PATH = os.path.join(PKG_DIR, "lib", "libdashinfer_hip_ops.so")

__all__ = ["PKG_DIR", "REPO_ROOT", "LIB_PATH", "OPS_LIB_PATH"]
