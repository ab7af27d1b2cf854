"""bench.py prints ONE JSON line with the fields the measurement contract names (a small corpus; a few seconds)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_has_the_contract_fields():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--docs", "40000", "--steps", "3", "--warmup", "1",
                          "--no-encoder", "--cpu-seconds", "1", "--min-seconds", "0.05", "--k", "100", "--built-docs", "3000",
                          "--built-kmeans-iters", "3", "--built-1m-docs", "4000"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["ranks_seen"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["value"] > 0 and abs(d["value"] - 32 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-3
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "ms_per_launch", "units_per_launch"):
        assert key in r, key
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None                       # PMC bytes are only reported for the workload they were measured on
    # the copy rate is MEASURED in the run (clb_measure_copy_rate: 5 x 1 GB device-to-device), not a constant of the script
    if r["bound"] == "hbm":
        assert 1000.0 < r["measured_copy_rate"] < 8000.0 and "clb_measure_copy_rate" in r["measured_copy_rate_how"]
        assert abs(r["frac_of_measured_copy_rate"] - r["achieved"] / r["measured_copy_rate"]) < 1e-3
        assert 1000.0 < r["measured_read_rate"] < 8000.0
    # the scalars a reader of the first 2 KB of the line needs sit in `config` (the driver's record keeps config, not the sub-records)
    cfgd = d["config"]
    for key in ("batch", "exchange", "p50_latency_ms", "p50_text_to_topk_ms", "end_to_end_with_query_encoder_qps", "end_to_end_serving_shape_qps",
                "built_index_1M_qps", "uniform_codes_qps", "fixed_batch_32_qps", "single_exchange_qps"):
        assert key in cfgd, key
    assert cfgd["batch"] == 32 and cfgd["exchange"] is None and cfgd["p50_latency_ms"] == d["p50_latency_ms"]
    assert cfgd["built_index_1M_qps"] == d["built_index_1M"]["value"] and cfgd["uniform_codes_qps"] == d["worst_case_uniform_codes"]["value"]
    assert len(json.dumps({k_: d[k_] for k_ in list(d)[:list(d).index("config") + 1]})) < 2048      # ... and within the first 2 KB
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] == "port" and c["gpu_matches_cpu_top_k"] is True
    assert d["in_flight_matches_serial"] is True
    # the sub-records of the line: worst case (uniform codes), BASELINE config 2 on this repo's own index build, batch sweep
    w = d["worst_case_uniform_codes"]
    assert w["value"] > 0 and w["gpu_matches_cpu_top_k"] is True and "frac" in w["roofline"] and "uniform" in w["workload"].lower()
    b = d["built_index"]
    assert b["value"] > 0 and b["gpu_matches_cpu_top_k"] is True and "frac" in b["roofline"]
    ib = b["index_build"]
    for key in ("kmeans_s", "kmeans_iters", "codec_stats_s", "compress_s", "build_ivf_s", "kmeans_roofline", "K"):
        assert key in ib, key
    assert b["candidates_per_query"]["passages"] > 0
    # ... and the device-resident build (at a test size here; 1 M passages in the default run)
    m = d["built_index_1M"]
    assert m["value"] > 0 and m["gpu_matches_cpu_top_k"] is True and "frac" in m["roofline"]
    for key in ("kmeans_s", "kmeans_iters", "codec_stats_s", "compress_s", "build_ivf_s", "kmeans_roofline", "compress_roofline", "K",
                "sample_points", "total_build_s"):
        assert key in m["index_build"], key
    assert set(d["batch_sweep"]) == {"64", "128", "256"}
