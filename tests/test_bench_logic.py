"""CPU-only: the arithmetic bench.py applies to counters (no GPU, no rocprofv3): the corrected roofline.traffic and the committed stream breakdown it uses."""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    sys.path.insert(0, ROOT)
    return importlib.import_module("bench")


def test_committed_fetch_breakdown_is_consistent():
    bench = _bench()
    bd, name = bench.load_fetch_breakdown()
    assert bd and name.startswith("r") and name.endswith("_fetch_breakdown.json")
    fr, raw = bd["fractions_of_raw_fetch"], bd["raw_fetch_B_per_step"]
    assert abs(sum(fr.values()) - 1.0) < 1e-9 and all(0.0 < v < 1.0 for v in fr.values())
    assert abs(raw["probe"] + raw["chain"] + raw["giants"] - raw["total"]) < 1e-6
    # the experiment libraries that gave the split identify themselves as such, the shipped one does not
    f = bd["FETCH_SIZE"]
    assert f["shipped"]["library_build_info"] == "" and "WRONG-RESULTS" in f["no_chain"]["library_build_info"] and "WRONG-RESULTS" in f["giants_cached"]["library_build_info"]
    # the probe share is the algorithmic 64 bytes per step within 10 % (random lines are counted 1.00x)
    assert 57.0 < raw["probe"] / bd["calibration_ratios"]["random_64B_lines"] < 70.0
    assert 0.45 < bd["calibration_ratios"]["coalesced_16B_loads"] < 0.55 and 0.45 < bd["calibration_ratios"]["coalesced_16B_lds_dma"] < 0.55


def test_corrected_traffic_divides_every_stream_by_its_own_calibration_ratio():
    bench = _bench()
    bd, _ = bench.load_fetch_breakdown()
    fr = bd["fractions_of_raw_fetch"]
    steps = 192 << 25
    m = {"fetch_bytes_per_step": 72.0, "write_bytes_per_step": 4.0,
         "calibration_ratios": {"random_64B_lines": 1.0, "coalesced_16B_loads": 0.5, "coalesced_16B_lds_dma": 0.5, "nt_16B_stores": 1.0}}
    ct = bench.corrected_traffic(m, steps)
    want = 72.0 * fr["probe"] / 1.0 + 72.0 * fr["chain"] / 0.5 + 72.0 * fr["giants"] / 0.5 + 4.0
    assert abs(ct["bytes_per_step"] - want) < 1e-9 and abs(ct["bytes_per_launch"] - want * steps) < 1.0
    assert abs(sum(ct["fetch_breakdown_B_per_step"].values()) + ct["write_B_per_step"] - want) < 1e-9
    # ratios of 1 everywhere: the corrected figure is the raw one
    m["calibration_ratios"] = {k: 1.0 for k in m["calibration_ratios"]}
    assert abs(bench.corrected_traffic(m, steps)["bytes_per_step"] - 76.0) < 1e-9
    # no counters, no figure
    assert bench.corrected_traffic({"calibration_ratios": {}}, steps) is None
    json.dumps(ct)


def test_box_independent_figures():
    """value_per_GHz and nJ_per_giant_step: the same kernel on a faster-clocked box gives a higher rate and the same two figures; a kernel that
    needs more cycles or more energy per step moves them; ranks add up; missing samples give None, never a made-up number"""
    bench = _bench()
    a = bench.box_independent([{"giant_steps_per_s": 41.0e9, "sclk_MHz": 1740.0, "socket_W": 1370.0, "idle_W": 245.0}])
    b = bench.box_independent([{"giant_steps_per_s": 38.76e9, "sclk_MHz": 1645.0, "socket_W": 1308.5, "idle_W": 245.0}])      # the slow box of round 4: -5.5 % rate
    assert abs(a["value_per_GHz"] / b["value_per_GHz"] - 1.0) < 0.002 and abs(a["nJ_per_giant_step"] / b["nJ_per_giant_step"] - 1.0) < 0.002
    assert abs(a["nJ_per_giant_step"] - (1370.0 - 245.0) / 41.0) < 1e-9
    worse = bench.box_independent([{"giant_steps_per_s": 39.0e9, "sclk_MHz": 1740.0, "socket_W": 1370.0, "idle_W": 245.0}])           # 5 % more cycles per step at the same clock
    assert worse["value_per_GHz"] < 0.96 * a["value_per_GHz"] and worse["nJ_per_giant_step"] > 1.04 * a["nJ_per_giant_step"]
    two = bench.box_independent([{"giant_steps_per_s": 41.0e9, "sclk_MHz": 1740.0, "socket_W": 1370.0, "idle_W": None}] * 2)
    assert abs(two["value_per_GHz"] - 2 * a["value_per_GHz"]) < 1.0 and abs(two["nJ_per_giant_step"] - a["nJ_per_giant_step"]) < 1e-9      # idle defaults to 245 W
    none = bench.box_independent([{"giant_steps_per_s": 41.0e9, "sclk_MHz": None, "socket_W": None, "idle_W": None}])
    assert none["value_per_GHz"] is None and none["nJ_per_giant_step"] is None
