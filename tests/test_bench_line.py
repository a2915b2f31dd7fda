"""The stdout line of bench.py is read by a parser with a size limit (round 4's 23 KB line came back unparsed): the line must
stay short, keep the contract's keys, and point at the detail file that holds the full objects."""
import importlib.util
import json
import os
import sys

import util

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(util.REPO, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


def test_stdout_line_is_short_and_keeps_the_contract_keys():
    b = _bench()
    full = json.loads(open(os.path.join(util.REPO, "profiles", "r4_bench_f32.json")).read().strip().splitlines()[-1])
    assert len(json.dumps(full)) > 20000                      # the record that was too long for the driver
    line = b.compact_line(full, "gpurun_out/bench_detail_f32.json")
    text = json.dumps(line)
    assert len(text) < 6000, len(text)
    assert json.loads(text) == line                           # strict JSON: no NaN / Infinity tokens
    for k in CONTRACT:
        assert k in line, k
    assert line["value"] == float("%.6g" % full["value"]) and line["config"]["workload"].startswith("BASELINE configs[1]")
    r = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    s = line["summary"]
    for k in ("b64_f32_decoder_gemm_frac", "b170_f32_commits_per_s", "b64_bf16_commits_per_s", "spmm_b64_frac",
              "decode_tokens_per_s", "gcn_frac_hbm"):
        assert k in s, k
    assert line["detail"].endswith(".json")


def test_non_finite_numbers_never_reach_the_line():
    b = _bench()
    assert b._sig({"a": float("nan"), "b": [float("inf"), 1.23456789]}) == {"a": None, "b": [None, 1.23457]}
