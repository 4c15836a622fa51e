"""bench.py's bookkeeping that needs no GPU: which committed counter collection a bench line quotes (the newest of the
right variant), and the `counters_stale` flag that compares the collection's build id with the running library's
(VERDICT r03 weak 9 / ADVICE r03); the cached synthetic PCM; the dominance rule over HBM- and f64-priced kernels."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _bench(monkeypatch, tmp_path, build="aaaa"):
    import bench

    (tmp_path / "profiles").mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "running_build_id", lambda: build)
    return bench, tmp_path / "profiles"


def _write(pdir, name, build, kernel="k_cand64p", **extra):
    body = {"_build_id": build, kernel: {"hbm_bytes": 280e6, "fetch_bytes": 275e6, "write_bytes": 5e6, "SQ_INSTS_VALU": 81.4e6,
                                         "valu_active_per_wave_cycle": 0.33}}
    body.update(extra)
    (pdir / name).write_text(json.dumps(body))


def test_newest_collection_of_the_right_variant_and_stale_flag(monkeypatch, tmp_path):
    bench, pdir = _bench(monkeypatch, tmp_path, build="aaaa")
    assert bench.profile_figures(3, "k_cand64", 0.17) == (None, None, None, None)          # nothing collected
    _write(pdir, "r03_f_traffic.json", None)                                                 # a collection without a build id
    _write(pdir, "r04_cfg5_traffic.json", "aaaa")                                            # another configuration's
    _write(pdir, "r04_hi_traffic.json", "aaaa")                                              # another signal's
    traffic, src, valu, stale = bench.profile_figures(3, "k_cand64", 0.17)
    assert traffic == 280e6 and src["source"] == "profiles/r03_f_traffic.json" and valu is None and stale is True
    _write(pdir, "r04_b_traffic.json", "aaaa")                                               # newer, of this build
    _write(pdir, "r04_b_valu.json", "aaaa")
    traffic, src, valu, stale = bench.profile_figures(3, "k_cand64", 0.17)
    assert src["source"] == "profiles/r04_b_traffic.json" and valu["source"] == "profiles/r04_b_valu.json" and stale is False
    _write(pdir, "r04_b_valu.json", "bbbb")                                                  # counters of another build
    assert bench.profile_figures(3, "k_cand64", 0.17)[3] is True
    # variants are kept apart: the high-order input of config 3, config 5, config 5 on its high-order input
    assert bench.profile_figures(3, "k_cand64", 0.2, "hi")[1]["source"] == "profiles/r04_hi_traffic.json"
    assert bench.profile_figures(5, "k_cand64", 0.2)[1]["source"] == "profiles/r04_cfg5_traffic.json"
    assert bench.profile_figures(5, "k_cand64", 0.3, "hi")[1] is None
    # 5..8 channels: the frame assembly is k_sub64
    _write(pdir, "r04_cfg4_traffic.json", "aaaa", kernel="k_sub64")
    assert bench.profile_figures(4, "k_pack", 0.5)[1]["kernel"] == "k_sub64"


def test_issue_floor_joins_with_its_own_build_id(monkeypatch, tmp_path):
    bench, pdir = _bench(monkeypatch, tmp_path, build="aaaa")
    _write(pdir, "r04_b_traffic.json", "aaaa")
    _write(pdir, "r04_b_valu.json", "aaaa")
    (pdir / "r04_issue_floor.json").write_text(json.dumps({"_build_id": "zzzz", "config3": {"k_cand64": {"attainable_ms": 0.137}}}))
    _, _, valu, stale = bench.profile_figures(3, "k_cand64", 0.17)
    assert valu["attainable_ms"] == 0.137 and abs(valu["frac_of_attainable"] - 0.137 / 0.17) < 1e-3 and stale is True
    (pdir / "r04_issue_floor.json").write_text(json.dumps({"_build_id": "aaaa", "config3": {"k_cand64": {"attainable_ms": 0.137}}}))
    assert bench.profile_figures(3, "k_cand64", 0.17)[3] is False


def test_synthetic_pcm_is_cached_and_identical(monkeypatch, tmp_path):
    import bench

    monkeypatch.setenv("FLAC_BENCH_CACHE", str(tmp_path))
    a = bench.make_pcm(4242, 3, 2, 16)
    files = sorted(os.listdir(tmp_path))
    assert len(files) == 1 and files[0].startswith("flacbench_ar2_4242_2_16_")
    b = bench.make_pcm(4242, 3, 2, 16)                       # read back
    assert np.array_equal(a, b) and a.size == 3 * bench.BLOCK * 2
    c = bench.make_pcm(4242, 3, 2, 16, "hi", 6)              # another signal: another file
    assert len(os.listdir(tmp_path)) == 2 and not np.array_equal(a, c)
    (tmp_path / files[0]).write_bytes(b"garbage")            # a damaged cache file is regenerated, not trusted
    assert np.array_equal(bench.make_pcm(4242, 3, 2, 16), a)


def test_dominance_rule_prices_both_kinds():
    import bench

    kernels = {"k_autocorr": {"ms": 0.30, "GFLOP/s": 29400.0}, "k_cand64": {"ms": 0.17, "GB/s": 3138.0}}
    alg = {"k_autocorr": ("f64", 8.9e9), "k_cand64": ("hbm", 546e6)}
    r = bench.roofline_of(kernels, "k_autocorr", alg, None, None, None, None)
    assert r["bound"] == "f64-valu" and r["unit"] == "TFLOP/s" and abs(r["frac"] - 29400.0 / 78600.0) < 1e-4
    r = bench.roofline_of(kernels, "k_cand64", alg, 280e6, {"source": "x"}, None, False)
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["counters_stale"] is False and r["traffic"] == 280e6


def _canned_full_record():
    """The builder's full r04 record (21 KB as one line: the one the driver could not parse), the canned input of the
    compact-line test."""
    return json.load(open(os.path.join(ROOT, "profiles", "r04_bench.json")))


REQUIRED_TOP = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")
REQUIRED_ROOFLINE = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "counters_stale")
REQUIRED_CPU = ("value", "unit", "cores", "kind", "sample")


def test_final_line_is_compact_strict_json_with_every_contract_key():
    """BENCH_r04.json.parsed was null because bench.py printed one 21 KB line.  The last stdout line
    is a compact record: strict JSON, one line, < 8 KB (target < 4 KB), every contract key present."""
    import bench

    full = _canned_full_record()
    assert len(json.dumps(full)) > 16000                       # the canned record really is the oversized one
    line = bench.final_line(full, "gpurun_out/bench_detail.json")
    assert "\n" not in line and len(line.encode()) < bench.FINAL_LINE_LIMIT
    assert len(line.encode()) < bench.FINAL_LINE_TARGET, len(line)

    def no_constants(x):
        raise AssertionError(f"non-strict JSON constant {x}")

    rec = json.loads(line, parse_constant=no_constants)
    for k in REQUIRED_TOP:
        assert k in rec, k
    for k in REQUIRED_ROOFLINE:
        assert k in rec["roofline"], k
    for k in REQUIRED_CPU:
        assert k in rec["cpu_baseline"], k
    assert rec["config"]["workload"] and "model" not in rec["config"]
    assert rec["value"] == full["value"] and rec["ms_per_step"] == full["ms_per_step"]
    assert rec["roofline"]["frac"] == full["roofline"]["frac"]
    assert abs(rec["roofline"]["frac"] - rec["roofline"]["achieved"] / rec["roofline"]["peak"]) < 1e-3
    assert rec["cpu_baseline"]["reference_fork_join"]["value"] == full["cpu_baseline"]["reference_fork_join"]["value"]
    ex = rec["extra"]
    assert ex["build_id"] == full["build_id"] and ex["sustained_ms_per_step"] == full["sustained"]["ms_per_step"]
    assert ex["high_order"]["ms_per_step"] == full["variants"]["high_order_input"]["ms_per_step"]
    for c in (2, 4, 5):
        assert ex[f"config{c}"]["ms_per_step"] == full["other_configs"][f"config{c}"]["ms_per_step"]
    assert ex["parity"]["identical_per_rank"] == ex["parity"]["checked_per_rank"] > 0
    # no prose beyond the workload / sample strings
    assert not any(isinstance(v, str) and len(v) > 64 for v in ex.values())


def test_final_line_survives_missing_blocks_and_oversized_extras():
    import bench

    full = _canned_full_record()
    lean = {k: v for k, v in full.items() if k not in ("other_configs", "end_to_end", "variants", "sustained", "cpu_baseline")}
    rec = json.loads(bench.final_line(lean))
    assert rec["cpu_baseline"] is None and rec["roofline"]["kernel"]         # --no-cpu-baseline runs, N > 1 ranks
    # a future block that blows the extras up is dropped, the contract keys stay
    full["kernels"] = {f"k_{i}": {"ms": i * 0.001} for i in range(600)}
    line = bench.final_line(full)
    assert len(line.encode()) < bench.FINAL_LINE_TARGET
    rec = json.loads(line)
    assert "kernel_ms" not in rec["extra"] and rec["roofline"] and rec["cpu_baseline"]


def test_detail_file_holds_the_full_record(tmp_path):
    import bench

    full = _canned_full_record()
    path = bench.write_detail(full, str(tmp_path / "detail.json"))
    assert json.load(open(path)) == full
