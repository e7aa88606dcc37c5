"""bench.py's last stdout line (CPU): the compact record built from a full record must stay far below what the driver's capture holds
(round-5 VERDICT: a 31.6 KB line left BENCH_r05.parsed null) and must carry the contract's fields."""
import io
import json
import os
import sys
from contextlib import redirect_stderr, redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CANNED = os.path.join(ROOT, "profiles", "round5_bench_line.json")       # a full record as round 5 printed it (31 KB)


def _detail():
    d = json.load(open(CANNED))
    d["box"] = {"gpu": "AMD Instinct MI355X", "cus": 256, "sclk_max_mhz": 2400, "mclk_max_mhz": 2000, "power_cap_w": 1400, "perf_level": "auto",
                "sclk_mhz_idle": 132, "sclk_mhz_load": 1540, "mclk_mhz_load": 2000, "power_w_load": 1390.0, "samples": 120}
    d["roofline"]["probe_steps"] = 5
    d["cpu_baseline"]["steps_b2"] = 4
    return d


def test_compact_line_is_small_and_complete():
    d = _detail()
    assert len(json.dumps(d)) > 20000                    # the canned record is the one that broke the driver's capture
    line = bench.compact_line(d)
    text = json.dumps(line)
    assert len(text) < bench.COMPACT_LIMIT, len(text)
    assert len(text) < 3200, len(text)                   # the target of the verdict (<= 3 KB) with room for longer box strings
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline", "sustained", "parity_mode", "batch32", "fp16", "channel_factor2", "channel_factor0.5"):
        assert key in line, key
    assert line["value"] == d["value"] and line["ms_per_step"] == d["ms_per_step"]
    assert set(line["config"]) == {"workload", "global_batch", "parallelism", "launch"}
    rf = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "step_frac", "step_frac_reference_flops", "conv_frac", "sn3x3_bwd_frac",
                "nonconv_floor_ms", "dominant_kernel"):
        assert rf.get(key) is not None, key
    assert rf["frac"] == d["roofline"]["frac"] and rf["sn3x3_bwd_frac"] == d["roofline"]["sn3x3_bwd"]["frac"]
    assert set(rf["dominant_kernel"]) == {"name", "launches", "avg_us", "gflop"}
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 16 and cb["unit"] == "images/sec" and cb["batch20"] == d["cpu_baseline"]["batch20"]["value"]
    assert isinstance(cb["sample"], str) and len(cb["sample"]) < 200
    assert isinstance(line["fp16"], float) and isinstance(line["channel_factor0.5"], float)      # one number per sub-record
    assert line["parity"]["fp32_b20_pixels"] == d["parity_b20"]["worst_pixel_abs_err"]
    assert line["detail"] == bench.DETAIL_FILE


def test_emit_prints_the_compact_line_last(tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    out, err = io.StringIO(), io.StringIO()
    with redirect_stdout(out), redirect_stderr(err):
        bench.emit(_detail())
    lines = out.getvalue().strip().splitlines()
    assert len(lines) == 1 and len(lines[0]) < bench.COMPACT_LIMIT
    parsed = json.loads(lines[-1])
    assert parsed["roofline"]["frac"] is not None and parsed["cpu_baseline"]["value"] is not None
    full = json.load(open(os.path.join(str(tmp_path), bench.DETAIL_FILE)))
    assert "routes" in full["roofline"] and json.loads(err.getvalue())["value"] == parsed["value"]


def test_multi_gpu_line_is_small():
    d = _detail()
    d["n_gpus"] = 8
    d["config"]["collectives"] = "nccl"
    d["multi_gpu"] = {"per_rank_images_per_sec": [1301.25] * 8, "allreduce_ms_d": 0.91, "allreduce_ms_g": 1.62, "exposed_ms": 0.4,
                      "exposed_ms_d": 0.0, "exposed_ms_g": 0.4, "grad_bytes_d": 67300000, "grad_bytes_g": 119900000, "note": "x" * 400}
    line = bench.compact_line(d)
    assert len(json.dumps(line)) < bench.COMPACT_LIMIT
    assert line["multi_gpu"]["per_rank"] == [1301.25] * 8 and line["config"]["collectives"] == "nccl"


def test_dpm_level_parsing():
    levels, cur = bench._dpm_levels("0: 132Mhz \n1: 1700Mhz *\n2: 2400Mhz \n")
    assert levels == [132, 1700, 2400] and cur == 1700
    assert bench._dpm_levels(None) == ([], None)
    assert bench.gpu_sample(None) == {"sclk_mhz": None, "mclk_mhz": None, "power_w": None}


def test_dominant_route_rule():
    assert bench.is_dominant_route("conv3x3_pp<16bit,2,FAST>") and bench.is_dominant_route("conv3x3_pp<16bit,2>")
    assert not bench.is_dominant_route("conv3x3_pp<16bit,2,FAST,w16>") and not bench.is_dominant_route("conv3x3_pp<16bit,1,FAST>")
    assert not bench.is_dominant_route("conv3x3_ppw<16bit> (64 co x 4 rows per wave)")
    assert bench.is_dominant_route("conv3x3_tall<f32,2,8>")
