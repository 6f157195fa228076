"""Runs one peer-to-peer loop-back scenario of tests/tp_loopback_lib.py in a process of its own (see
test_gpu_tp_loopback.py::run_worker).  usage: p2p_worker.py allreduce <nranks> | decode <nranks> <kv> <batch> <wbits> <group> <overlap> | hostdecode <nranks> <kv> <batch> <wbits> <group>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import _load_pkg  # noqa: E402

_load_pkg()
from tests import tp_loopback_lib  # noqa: E402

if sys.argv[1] == "allreduce":
    tp_loopback_lib.run_p2p_allreduce(int(sys.argv[2]))
elif sys.argv[1] == "hostfile":  # hostfile <nranks> <kv> <batch> <wbits> <group> <n_kv> <file|bound> <out.npz>: weights split at load from ONE export / bound as tensors
    tp_loopback_lib.run_tp_decode_host_from_file(int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]),
                                                 sys.argv[8], sys.argv[9])
elif sys.argv[1] == "hostdecode":  # hostdecode <nranks> <kv> <batch> <wbits> <group>: the C++ operator layer, a rank per thread
    tp_loopback_lib.run_tp_decode_host(int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]),
                                       graph=len(sys.argv) > 7 and sys.argv[7] == "graph")
else:
    nranks, kv, batch, wbits, group, overlap = int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), bool(int(sys.argv[7]))
    tp_loopback_lib.run_tp_decode(nranks, kv, batch, wbits, group, sys.argv[8] if len(sys.argv) > 8 else "p2p", overlap)
print("P2P_WORKER_OK")
