"""bench.py's launch plumbing on CPU: `python bench.py --gpus N` launches its own ranks (torch.distributed.run on 127.0.0.1), a box with
fewer GPUs than ranks gets ONE JSON error line and a non-zero return code (no traceback), and the world-size-2 path over gloo times
blocks with barrier + max-over-ranks and prints one line from rank 0."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(argv, env=None, timeout=240):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None), e.pop("RANK", None), e.pop("LOCAL_RANK", None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + argv, capture_output=True, text=True, timeout=timeout, env=e, cwd=ROOT)


def test_too_few_gpus_is_a_json_error_line_not_an_assert():
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    p = _run(["--gpus", str(have + 2)])
    assert p.returncode == 2, (p.returncode, p.stderr[-400:])
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["error"] == f"needs {have + 2} GPUs, {have} visible" and d["value"] is None and d["n_gpus"] == have + 2
    assert "Traceback" not in p.stderr and "AssertionError" not in p.stderr


def test_world_size_mismatch_under_a_launcher_is_an_error_line():
    p = _run(["--gpus", "2", "--workload", "launch_selftest"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    # WORLD_SIZE is set (a launcher started us) and disagrees with --gpus
    assert p.returncode == 2 and json.loads(p.stdout.splitlines()[-1])["error"] == "WORLD_SIZE 1 != --gpus 2"
    assert "Traceback" not in p.stderr


def test_self_launch_world_2_over_gloo_prints_one_line_with_the_slowest_rank():
    p = _run(["--gpus", "2", "--workload", "launch_selftest", "--steps", "5", "--warmup", "1"], env={"DIHIP_BENCH_BACKEND": "gloo"})
    assert p.returncode == 0, p.stderr[-600:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["comm_backend"] == "gloo"
    # rank r sleeps 2 (1 + r) ms per step: the line carries the MAXIMUM over the ranks (rank 1: >= 4 ms per step)
    assert d["ms_per_step"] >= 3.9, d
