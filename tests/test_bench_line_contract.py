"""The committed driver-style bench line (profiles/r06zi_bench_default.json: the stdout of `python bench.py --gpus 1 --steps 20 --warmup 5`, the
driver's own command, on an MI355X) against the contract the driver reads: ONE JSON line under 4 KB with metric / value / unit / n_gpus / steps /
warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload, a `roofline` object for the TIME-dominant
kernel (+ `roofline_gemv`), a `cpu_baseline` object -- and internally consistent numbers (value = batch / ms_per_step, roofline.frac = achieved /
peak, achieved = algorithmic bytes / the profiler's average duration, the step's HBM fraction from its algorithmic bytes).  Round 5's line was
21 KB and the driver could not parse it (BENCH_r05.json: parsed null): the size is part of the contract now.  Guards the SHAPE of the line on
CPU; the numbers themselves are the GPU box's.  The full record of round 5 (profiles/r05z_bench_default.json) feeds headline_line() below."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    """round 5's full 21 KB record: everything a run can attach (input of headline_line)"""
    return json.load(open(os.path.join(ROOT, "profiles", "r05z_bench_default.json")))


def _stdout():
    return open(os.path.join(ROOT, "profiles", "r06zi_bench_default.json")).read()


def _headline():
    lines = [l for l in _stdout().splitlines() if l.strip()]
    return json.loads(lines[-1])


def _second(d):
    """the runner-up roofline object, named by what it is (`roofline_attn_block` / `roofline_gemv`)"""
    return d.get("roofline_attn_block") or d["roofline_gemv"]


def test_stdout_is_one_small_line():
    lines = [l for l in _stdout().splitlines() if l.strip()]
    assert len(lines) == 1 and len(lines[0]) < 4096, (len(lines), len(lines[-1]))


def test_required_fields_and_types():
    d = _headline()
    for k, t in (("metric", str), ("value", (int, float)), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", (int, float)),
                 ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert k in d and isinstance(d[k], t), k
    assert "vs_baseline" in d and d["vs_baseline"] is None          # BASELINE.md holds no published number for this metric
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] in ("weak", "strong") and d["unit"] == "tokens/s"
    assert d["steps"] == 20 and d["warmup"] == 5                     # the driver's flags
    assert "synthetic" in d["data"] and "workload" in d["config"] and "model" not in d["config"]
    for r in (d["roofline"], _second(d)):
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
            assert k in r, k
        assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] == 8000.0
        assert r["traffic"] is not None and r["traffic"] >= r["algorithmic_bytes_per_launch"]      # PMC bytes per launch, never below the algorithmic ones
    # `roofline` is the kernel the step spends most of its time in; the runner-up rides along: the two are the fused attention block and the
    # RMSNorm + gate/up + SwiGLU GEMV (since the second half of round 6 the GEMV leads: the block went from 16.7 to 14.3 us)
    assert d["roofline"]["share_of_step"] >= _second(d)["share_of_step"]
    names = d["roofline"]["kernel"] + " | " + _second(d)["kernel"]
    assert "decode_attn_block_kernel" in names and "gate/up" in names
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["unit"] == d["unit"]


def test_numbers_are_consistent():
    d = _headline()
    batch = d["config"]["global_batch"]
    assert abs(d["value"] - batch * 1e3 / d["ms_per_step"]) <= 2e-3 * d["value"]
    for r in (d["roofline"], _second(d)):
        assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 2e-4
        # achieved = algorithmic bytes per launch / the kernel's average duration on the profiler's clock
        assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_kernel_us_rocprof"] * 1e-6) / 1e9) <= 2e-3 * r["achieved"]
        assert abs(r["share_of_step"] - r["avg_kernel_us_rocprof"] * d["config"]["layers"] / (d["ms_per_step"] * 1e3)) <= 2e-3
    h = d["step_hbm"]
    assert abs(h["frac_of_peak"] - h["achieved_GBps_per_gpu"] / 8000.0) <= 2e-4
    assert abs(h["achieved_GBps_per_gpu"] - h["algorithmic_bytes_per_rank"] / (d["ms_per_step"] * 1e-3) / 1e9) <= 2e-3 * h["achieved_GBps_per_gpu"]
    b = d["blocks"]
    assert b["count"] == 5 and b["ms_per_step_min"] <= d["ms_per_step"] <= b["ms_per_step_max"]
    # `value` is the C++ operator layer's figure; the Python runner's is beside it
    assert d["runner"].startswith("host") and d["python_runner_tokens_per_s"] > 0
    # the committed rocprofv3 summary of the same command agrees with the line's kernel durations (profiles/r06zf_bench_int4_b1_kernel_stats.csv)
    import csv
    rows = {r["Name"]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(os.path.join(ROOT, "profiles", "r06zf_bench_int4_b1_kernel_stats.csv")))}
    blk = [v for k, v in rows.items() if "decode_attn_block_kernel" in k][0]
    in_line = [r for r in (d["roofline"], _second(d)) if "decode_attn_block_kernel" in r["kernel"]][0]
    assert abs(blk - in_line["avg_kernel_us_rocprof"]) <= 0.05 * blk


def test_extra_workloads_are_the_named_ones():
    d = _headline()
    assert [w.get("workload") for w in d["extra"]] == ["int4_b32_u4kv", "int8_b1", "prefill_2048", "cfg3_rank", "tp8_rank_7b", "cfg5_moe"]
    for w in d["extra"]:
        assert "error" not in w and "skipped" not in w, w
        assert w["value"] > 0 and w["ms_per_step"] > 0 and w["roofline_frac"] > 0
    assert d["detail"] == "gpurun_out/bench_detail.json" and d["wall_s"] < 420


# ---- round 6: the line the driver parses is SMALL and LAST (BENCH_r05.json came back `parsed: null` on a 21 KB line) ---------------------
def _bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_headline_line_of_a_full_record_is_under_4_kb_and_keeps_the_contract():
    b = _bench()
    full = _line()                                     # round 5's 21 KB record: everything a run can attach
    line = b.headline_line(full, "gpurun_out/bench_detail.json")
    text = json.dumps(line)
    assert len(text) < 4096, len(text)
    assert json.loads(text) == line
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "step_hbm", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["value"] == full["value"] and line["ms_per_step"] == full["ms_per_step"] and "workload" in line["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert k in line["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    # the secondary workloads ride as numbers only
    assert [w["workload"] for w in line["extra"]] == ["int4_b32_u4kv", "int8_b1", "prefill_2048", "cfg3_rank", "tp8_rank_7b", "cfg5_moe"]
    assert all(len(json.dumps(w)) < 260 for w in line["extra"])


def test_headline_line_sheds_optional_parts_before_it_exceeds_the_limit():
    b = _bench()
    full = _line()
    full["extra"]["workloads"] = full["extra"]["workloads"] * 8     # 48 secondary workloads: cannot fit
    full["tp_ab"] = [{"allreduce": "rccl", "overlap": False, "tokens_per_s": 1.0, "error": "x" * 500}] * 8
    line = b.headline_line(full, None)
    assert len(json.dumps(line)) < 4096
    for k in ("metric", "value", "ms_per_step", "config", "step_hbm", "roofline", "cpu_baseline"):
        assert k in line, k


def test_bench_prints_the_headline_last_and_alone():
    """Structure of main(): exactly one print of the headline on the rank-0 path, after the detail file is written."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    assert main.count("print(json.dumps(headline_line(out, detail)), flush=True)") == 1
    tail = main[main.index("print(json.dumps(headline_line(out, detail)), flush=True)"):]
    assert "print(" not in tail[len("print(json.dumps(headline_line(out, detail)), flush=True)"):]
