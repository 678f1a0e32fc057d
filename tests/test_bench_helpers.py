"""CPU: the pieces of bench.py that decide what the JSON line may claim."""
import json
import os

import numpy as np

import bench


def test_accuracy_object_definition():
    """ATE = sqrt(mean |dt|^2) without alignment, rotation = max 2 acos|q.q'| (diagnostics.py:114,122)"""
    n = 50
    ref = np.zeros((n, 16)); ref[:, 0] = 1.0
    ref[:, 4] = np.arange(n)
    got = ref.copy()
    got[:, 5] += 3e-7                                        # 0.3 um sideways everywhere
    a = 1e-3
    got[7, :4] = [np.cos(a / 2), 0, 0, np.sin(a / 2)]        # one keyframe yawed by 1 mrad
    acc = bench.accuracy_vs_oracle(got, dict(states=ref, gt=ref, updates=12))
    assert abs(acc["ate_m"] - 3e-7) < 1e-15 and abs(acc["rot_rad"] - a) < 1e-9
    assert acc["within_bar"] is False and acc["updates"] == 12 and acc["bar_m"] == 1e-6     # the rotation is over the bar
    acc = bench.accuracy_vs_oracle(ref + 0.0, dict(states=ref, gt=ref, updates=1))
    assert acc["within_bar"] is True and acc["ate_m"] == 0.0


def test_stale_traffic_file_is_refused(tmp_path, monkeypatch):
    """roofline.traffic comes from a committed PMC summary: it must be of the same profiling round as the kernel trace"""
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (prof / "kernel_durations.json").write_text(json.dumps({"source": "profiles/r03a_kernel_stats.csv + r03a_pmc_summary.md", "kernels": {}}))
    (prof / "traffic.json").write_text(json.dumps({"k1_bytes_per_imu_factor": 4416.0, "source": "profiles/r02e_pmc_summary.md (...)"}))
    t = bench.measured_traffic_per_imu_factor()
    assert "stale" in t and "r02e" in t["stale"] and "r03a" in t["stale"]
    (prof / "traffic.json").write_text(json.dumps({"k1_bytes_per_imu_factor": 4416.0, "source": "profiles/r03a_pmc_summary.md (...)"}))
    assert bench.measured_traffic_per_imu_factor()["k1_bytes_per_imu_factor"] == 4416.0


def test_committed_profiles_are_of_one_round():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    t = bench.measured_traffic_per_imu_factor()
    assert t is not None and "stale" not in t, t
    k = json.load(open(os.path.join(root, "profiles", "kernel_durations.json")))
    # the band solve of the profiled run: the assembling forward sweep (the default from 1 024 windows on) or the one-kernel form
    solve = next(n for n in ("vf::k_band_forward_asm2", "vf::k_band_forward_asm", "vf::k_band_solve") if n in k["kernels"])
    assert "sq" in k["kernels"][solve] and 0.0 < k["kernels"][solve]["sq"]["valu_issue_frac"] < 1.0


def test_cpu_leg_is_the_marginalised_update(oracle):
    """_cpu_updates: the update the GPU's timed step does -- its states after s updates equal a
    hand-rolled marginalise / predict / LM loop, and differ from the old re-anchoring loop."""
    from tests import helpers
    from vil_sensor_fusion_amd import synth
    n, steps, K = 60, 4, 4
    dt, snap, gt = bench._cpu_updates((3, n, n + steps + 1, steps, K, 1, steps, 30))
    seq = synth.make_sequence(seed=3, n_kf=n + steps + 1)
    prob = helpers.build_problem(oracle, seq)
    ref = helpers.FixedLagOracle(oracle, prob, n, K, init_iterations=30, ingest=(seq, oracle.carla_imu_params()))
    for _ in range(steps):
        ref.update()
    np.testing.assert_array_equal(snap, ref.window_states)
    # the appended factors were preintegrated again with the bias estimate of the moment (GraphManager.cpp:59), not with 0
    assert np.abs(ref.prob["imu"][n + steps - 1][10:16]).max() > 0 and not np.abs(prob["imu"][n + steps - 1][10:16]).any()
    assert ref.marg is not None and ref.marg.on == 1 and dt > 0 and gt.shape == (n, 16)


def test_the_one_json_line_fits_the_drivers_tail():
    """BENCH_r05.parsed was null because the line had grown to 21 KB.  The line is built from the full report
    (here: the committed 21 KB report of round 5) and must stay under 4 KB, parse, and carry the contract keys + roofline +
    cpu_baseline; everything else goes to bench_detail.json."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    full = json.load(open(os.path.join(root, "profiles", "r05d_bench_default.json")))
    assert len(json.dumps(full)) > 15000
    line = bench.compact_line(full)
    assert len(line) < 4096 and "\n" not in line
    got = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "detail"):
        assert k in got, k
    assert got["vs_baseline"] is None and got["config"]["workload"] and "model" not in got["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in got["roofline"], k
    assert abs(got["roofline"]["frac"] - got["roofline"]["achieved"] / got["roofline"]["peak"]) < 1e-5
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in got["cpu_baseline"], k
    assert abs(got["value"] / full["value"] - 1) < 1e-5 and abs(got["ms_per_step"] / full["ms_per_step"] - 1) < 1e-5
    assert "profiled_kernels" not in got and "issue_view" not in json.dumps(got)
    # a report stuffed with junk still yields a line that fits (optional objects are shed, the contract keys never)
    fat = dict(full, stage_ms={f"k{i}": 1.0 / 3 for i in range(400)})
    got = json.loads(bench.compact_line(fat))
    assert "roofline" in got and "cpu_baseline" in got and "stage_ms" not in got


def test_fallback_line_of_a_stalled_multi_rank_run_carries_roofline():
    fb = {"metric": "m", "value": 1.0, "unit": "keyframes/s", "n_gpus": 8, "steps": 2, "warmup": 1, "ms_per_step": 3.0,
          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
          "config": {"workload": "w"}, "roofline": {"bound": "hbm", "achieved": 5000.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.625, "traffic": None},
          "time_sharded_window": {"error": "no result"}}
    got = json.loads(bench.compact_line(fb))
    assert got["roofline"]["frac"] == 0.625 and got["roofline"]["traffic"] is None and got["time_sharded_window"]["error"]


def test_detail_file_is_written_where_asked(tmp_path, monkeypatch):
    monkeypatch.setenv("VF_BENCH_DETAIL_DIR", str(tmp_path))
    p = bench.write_detail({"a": [1, 2, 3]})
    assert json.load(open(p)) == {"a": [1, 2, 3]}
