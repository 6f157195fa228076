"""The one-shot peer-to-peer all-reduce between PROCESSES (IPC handles, dihip_ipc_get_handle / _open_handle): N rank
processes share GPU 0, exchange their receive buffers over a gloo process group and all-reduce decode-sized rows through
decoder.P2PComm(guarded=True) -- the set-up bench.py's "auto" backend uses on a node.  What this covers that the
thread loop-back (test_gpu_tp_loopback.py) cannot: the IPC mapping of a peer process's buffer and system-scope visibility
of the flags between processes.  What it does not: xGMI (one GPU)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 4])
def test_p2p_allreduce_between_processes_on_one_gpu(pkg, world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GPU_MAX_HW_QUEUES="8")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "p2p_proc_worker.py"), str(r), str(world), str(port)],
                              env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
            out += "\n<timeout>"
        outs.append(out)
    text = "\n".join(o[-1500:] for o in outs)
    if any("P2P_UNAVAILABLE" in o for o in outs):
        # the guarded set-up did its job (every rank left together, nothing hung); say why the path is not available here
        pytest.skip("peer-to-peer path unavailable between processes on this box: " + next(o for o in outs if "P2P_UNAVAILABLE" in o)[-300:])
    assert all(p.returncode == 0 for p in procs) and all("P2P_PROC_OK" in o for o in outs), text
