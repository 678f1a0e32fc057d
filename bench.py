#!/usr/bin/env python3
"""Headline benchmark: keyframes/sec of fixed-lag smoother updates on 1000-pose windows.

One *step* = one fixed-lag update of every window in the batch: append one keyframe (its IMU
factor and between factor are already resident in HBM; its initial value comes from the IMU
prediction, GraphManager.cpp:152-160), marginalise the oldest one into a dense prior (Schur complement, K-marg), then K
Levenberg-Marquardt trials, each = linearise ALL factors of the window (K1+K2) -> block-banded
J^T J (K3) -> banded Cholesky solve (K4) -> retract + cost + accept/reject (K5).
value = (windows on all ranks) * steps / max-over-ranks time: one new keyframe per window per step.

Contract: python bench.py --gpus N --steps K --warmup W ; rank 0 prints ONE JSON line.
N > 1: launched by torch.distributed.run, one rank per GPU; windows are independent, so ranks
share nothing on the data path ("scaling": "weak", windows per GPU fixed).

Beside the contract fields the line carries, all measured outside the timed region of `value`:
  roofline               K1 (the Jacobian kernel) from HIP events on the engine's stream, PMC traffic from profiles/
  roofline_solve         the same for K4, the kernel that takes most of a step
  stage_ms               one launch of every hot-path kernel
  with_convergence_exit  the same update with GTSAM's LM termination rule on (windows stop taking trials)
  time_sharded_window    ONE 10 000-pose window spread in time over all ranks (BASELINE configs[4])
  single_window          latency of the update for one window (what one vehicle sees)
  degeneracy_k6          the 6x6 degeneracy metrics, f64 / f32, beside the reference's per-matrix numpy calls (N = 1)
  graph_manager          latency of GraphManager::solve (the drop-in call) at a 1000-keyframe lag (N = 1)
  cpu_baseline(_openmp)  the C oracle doing the same update on the host (N = 1)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Algorithmic bytes of K1 per IMU factor: 1776 B read (190-double record + two 16-double states) + the residual and the
# 291 entries of the whitened 15x30 Jacobian that are not structurally zero, (15 + 291) * 8 = 2448 B written.
# SURVEY 8(d) quotes 5496 B for the dense block (465 * 8 = 3720 B written); 159 of its 450 entries are structural zeros
# which k_linearize_imu no longer writes, so the roofline is priced on the smaller figure; the dense one is kept beside it.
IMU_BYTES = 1776 + 8 * (15 + 450 - 159)
IMU_BYTES_DENSE = 5496
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def make_engine(args, rank, local_rank, windows):
    """Synthetic Carla-like factors for `windows` windows, preintegrated ON THE DEVICE (K0)."""
    from vil_sensor_fusion_amd import Engine, EngineOpts, synth
    from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS
    n, total = args.window, args.window + args.steps + args.warmup + 1
    eng = Engine(EngineOpts(windows=windows, capacity=total, device=local_rank))
    nseq = max(1, min(args.sequences, windows))
    seqs = [synth.make_sequence(seed=1000 * rank + s, n_kf=total) for s in range(nseq)]
    recs = [synth.between_records(s) for s in seqs]
    for w in range(windows):
        seq = seqs[w % nseq]
        gt0 = seq.gt_states[0]
        eng.preintegrate(w, 1, seq.imu_off[1:], seq.imu_steps, np.zeros(6), synth.CARLA_IMU_COV)
        eng.set_between(w, seq.btw_a, seq.btw_b, recs[w % nseq])
        eng.set_states(w, 0, gt0.reshape(1, 16))
        eng.set_prior(w, 0, synth.prior_record(gt0, REFERENCE_PRIOR_SIGMAS))
        eng.set_range(w, 0, 1)
    eng.predict(-1, 1, n - 1)         # initial values by IMU prediction, all windows in one launch
    for w in range(windows):
        eng.set_range(w, 0, n)
    eng.sync()
    eng.iterate(args.iterations)      # converge the initial windows (not timed)
    eng.sync()
    return eng


def cpu_baseline(args, threads=1):
    """The CPU oracle (a port, not GTSAM: GTSAM cannot be built here) doing the same update on a
    bounded sample: one window, `cpu_steps` updates of K LM trials each, on `threads` host cores
    (OpenMP over the factors of the linearisation; the banded solve is scalar)."""
    from oracle import oracle
    from tests import helpers
    from vil_sensor_fusion_amd import synth
    oracle.build()
    n = args.window
    seq = synth.make_sequence(seed=0, n_kf=n + args.cpu_steps)
    prob = helpers.build_problem(oracle, seq)
    t0 = time.perf_counter()
    for s in range(args.cpu_steps):
        win = helpers.oracle_window(oracle, prob, lo=s, hi=s + n)
        win.lm(iterations=args.iterations, n_threads=threads)
        prob["states"][s:s + n] = win.states
    dt = time.perf_counter() - t0
    return dict(value=args.cpu_steps / dt, unit="keyframes/s", cores=threads, kind="port",
                host_cores_available=os.cpu_count(),
                sample=f"{args.cpu_steps} fixed-lag updates of one {n}-pose window, {args.iterations} LM trials each, "
                       f"C restatement (oracle/vf_oracle.c) on {threads} thread(s); the reference's CPU GTSAM path cannot be "
                       f"built or timed here (no GTSAM/Eigen/Boost/ROS)")


def time_sharded_window(args, info, dist, backend, dev):
    """BASELINE.json configs[4]: ONE 10 000-pose window spread in time over the ranks (every rank owns
    96 / world chunks of the partitioned solve; separator blocks all-gathered, increments all-reduced
    over RCCL).  Outside the timed region of the headline metric; every rank takes part."""
    import torch
    from vil_sensor_fusion_amd import Engine, EngineOpts, synth, distributed as D
    from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS
    n, chunks, trials = args.sharded_window, 96, args.iterations
    eng, problem = None, None
    try:       # set-up has no collective in it: a rank that fails here must not leave the others waiting in one
        seq = synth.make_sequence(seed=4242, n_kf=n)
        eng = Engine(EngineOpts(windows=1, capacity=n + 8, device=dev.index, chunks=chunks))
        eng.preintegrate(0, 1, seq.imu_off[1:], seq.imu_steps, np.zeros(6), synth.CARLA_IMU_COV)
        eng.set_between(0, seq.btw_a, seq.btw_b, synth.between_records(seq))
        eng.set_states(0, 0, seq.gt_states[:1])
        eng.set_prior(0, 0, synth.prior_record(seq.gt_states[0], REFERENCE_PRIOR_SIGMAS))
        eng.set_range(0, 0, 1)
        eng.predict(0, 1, n - 1)
        eng.set_range(0, 0, n)
        eng.sync()
    except Exception as exc:   # noqa: BLE001
        problem = f"{type(exc).__name__}: {exc}"
    healthy = D.max_over_ranks(dist, 0.0 if problem is None else 1.0,
                               device=dev if (dist is not None and backend == "nccl") else "cpu") == 0.0
    if not healthy:
        if eng is not None:
            eng.close()
        return {"error": problem or "set-up failed on another rank"}
    solver = D.ShardedSolver(eng, dist, dev, backend=backend)
    solver.iterate(trials)                       # converge + warm up (not timed)
    torch.cuda.synchronize(dev)
    D.barrier(dist)
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        solver.iterate(trials)
    torch.cuda.synchronize(dev)
    dt = D.max_over_ranks(dist, time.perf_counter() - t0, device=dev if (dist is not None and backend == "nccl") else "cpu")
    lm = eng.read_lm(0)
    per_trial = dt / (reps * (trials + 1e-30))
    out = {"window_keyframes": n, "ranks": info.world, "chunks": chunks, "chunks_per_rank": chunks // max(info.world, 1),
           "lm_trials_timed": reps * trials, "ms_per_lm_trial": per_trial * 1e3,
           "keyframe_relinearisations_per_s": n / per_trial,
           "exchange_doubles_per_trial": chunks * 2241 + n * 15 + 2,      # (27x28 + 27x28 + 27x27) per chunk, increments, cost
           "collectives_per_trial": "3 all-gather (separator blocks) + 2 all-reduce (increments, cost)",
           "backend": backend if dist is not None else "none", "final_cost": lm["cost"], "solve_failures": lm["solve_failures"]}
    eng.close()
    return out


def degeneracy_section():
    """K6 (BASELINE.json configs[3]): the degeneracy metrics of the reference's Python library on a batch of
    6x6 information matrices, float64 and float32, kernel time from HIP events inside the library; beside it the
    per-matrix numpy/LAPACK calls the reference makes (degeneracy_detection_functions.py:38-83), on a bounded sample."""
    from vil_sensor_fusion_amd import degeneracy as dg
    rng = np.random.default_rng(7)
    T = 200_000
    A = rng.normal(size=(T, 6, 6))
    mats = np.ascontiguousarray((A @ A.transpose(0, 2, 1) + 0.5 * np.eye(6)).transpose(1, 2, 0))
    out = {"matrices": T, "gpu_ns_per_matrix": {}, "algorithmic_bytes_per_matrix": {"f64": 296, "f32": 148}}
    for name in ("d_opt", "e_opt", "condition_number"):
        for dt, tag in ((np.float64, "f64"), (np.float32, "f32")):
            _, ms = dg.apply_degen_function(mats, None, "all", name, dtype=dt, reps=5)
            out["gpu_ns_per_matrix"][f"{name}/{tag}"] = ms * 1e6 / T
    sample = np.ascontiguousarray(mats[:, :, :2000].transpose(2, 0, 1))
    t0 = time.perf_counter()
    for m in sample:                                   # one LAPACK call per matrix, as the reference does
        np.exp(np.log(np.linalg.det(m)) / 6)
    t1 = time.perf_counter()
    for m in sample:
        np.linalg.eigvals(m).real.min()
    t2 = time.perf_counter()
    out["numpy_per_matrix_ns"] = {"d_opt": (t1 - t0) / len(sample) * 1e9, "e_opt": (t2 - t1) / len(sample) * 1e9,
                                  "sample": len(sample), "cores": 1}
    return out


def graph_manager_section(lag=1000, extra=120):
    """Latency of the drop-in surface itself: GraphManager.solve() (= vf_solve) per keyframe in fixed-lag mode, fed like
    the node (IMU at 200 Hz between keyframes, one reserveNode + solve per keyframe), once the window is full."""
    from vil_sensor_fusion_amd import GraphManager, synth
    nkf = lag + extra
    seq = synth.make_sequence(seed=3, n_kf=nkf + 2)
    gm = GraphManager(capacity=lag + 192, lag=lag, iterations=5)
    t, times = 0.0, []
    for k in range(1, nkf):
        for s in seq.imu_steps[seq.imu_off[k]:seq.imu_off[k + 1]]:
            t += s[0]
            gm.addIMUMeasurement(t, s[1:4], s[4:7])
        gm.reserveNode(t)
        t0 = time.perf_counter()
        gm.solve()
        times.append(time.perf_counter() - t0)
    steady = np.array(times[lag + 20:]) * 1e3
    gm.close()
    return {"lag_keyframes": lag, "lm_trials_per_solve": 5, "solve_ms_mean": float(steady.mean()),
            "solve_ms_p99": float(np.percentile(steady, 99)), "solves_timed": int(steady.size),
            "what": "vf_solve: K0 of the new IMU factor, prediction, marginalisation of the oldest keyframe, 5 LM trials, read-back"}


def measured_traffic_per_imu_factor():
    """HBM bytes per IMU factor of K1 from the committed PMC profile (separate --pmc passes,
    2*FETCH_SIZE + WRITE_SIZE, KiB units; tools/summarize_prof.py)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(path):
        return json.load(open(path))
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--window", type=int, default=1000, help="keyframes per window (BASELINE metric: 1000)")
    ap.add_argument("--windows", type=int, default=1024, help="independent windows per GPU")
    ap.add_argument("--sequences", type=int, default=8, help="distinct synthetic sequences per rank")
    ap.add_argument("--iterations", type=int, default=5, help="LM trials per update")
    ap.add_argument("--cpu-steps", type=int, default=128,
                    help="fixed-lag updates the CPU baseline is timed on (128 = about 10 s on one core)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-window", action="store_true")
    ap.add_argument("--reanchor", action="store_true", help="drop the oldest keyframe by re-anchoring tight priors instead of marginalising it")
    ap.add_argument("--sharded-window", type=int, default=10000, help="keyframes of the time-sharded window (BASELINE configs[4])")
    ap.add_argument("--no-sharded", action="store_true")
    ap.add_argument("--no-convergence-exit", action="store_true")
    ap.add_argument("--no-degeneracy", action="store_true")
    ap.add_argument("--no-graph-manager", action="store_true")
    args = ap.parse_args()

    import torch
    from vil_sensor_fusion_amd import distributed as D
    from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS
    info = D.rank_info()
    dist = None
    # VF_BENCH_BACKEND=gloo + VF_BENCH_SHARE_GPU=1 let two ranks share GPU 0 (control-plane smoke test
    # on a 1-GPU box); the driver's N-GPU runs use RCCL ("nccl") with one rank per GPU.
    backend = os.environ.get("VF_BENCH_BACKEND", "nccl")
    gpu = 0 if os.environ.get("VF_BENCH_SHARE_GPU") else info.local_rank
    if info.world > 1 or os.environ.get("VF_FORCE_DIST"):
        torch.cuda.set_device(gpu)
        dist = D.init(backend=backend, device_id=torch.device("cuda", gpu) if backend == "nccl" else None)
    dev = torch.device("cuda", gpu)

    eng = make_engine(args, info.rank, gpu, args.windows)

    def fence():
        D.barrier(dist)
        torch.cuda.synchronize(dev)
        eng.sync()

    def one_step(e):
        e.slide(REFERENCE_PRIOR_SIGMAS, marginalize=not args.reanchor)
        e.iterate(args.iterations)

    for _ in range(args.warmup):
        one_step(eng)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step(eng)
    fence()
    dt = D.max_over_ranks(dist, time.perf_counter() - t0, device=dev if (dist is not None and backend == "nccl") else "cpu")
    summaries = D.gather_summaries(dist, dict(rank=info.rank, keyframes=args.windows * args.steps))

    # Same update with GTSAM's LM termination rule switched on (vf_engine_set_convergence): a second,
    # clearly labelled number -- the headline above always runs all K trials on every window.
    conv = None
    if not args.no_convergence_exit:
        sample = range(0, args.windows, max(1, args.windows // 64) | 1)   # ~64 windows, odd stride: every synthetic sequence
        before = [eng.read_lm(w) for w in sample]
        eng.set_convergence(1e-5, 1e-5)
        for _ in range(args.warmup):
            one_step(eng)
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            one_step(eng)
        fence()
        dt_c = D.max_over_ranks(dist, time.perf_counter() - t1, device=dev if (dist is not None and backend == "nccl") else "cpu")
        after = [eng.read_lm(w) for w in sample]
        trials = [(a["accepted"] + a["rejected"] - b["accepted"] - b["rejected"]) / (args.steps + args.warmup) for a, b in zip(after, before)]
        eng.set_convergence(0.0, 0.0)
        conv = {"value": info.world * args.windows * args.steps / dt_c, "unit": "keyframes/s", "ms_per_step": dt_c / args.steps * 1e3,
                "rule": "a window stops after a trial that changes its cost by <= 1e-5 absolute or relative "
                        "(gtsam LevenbergMarquardtParams defaults), at most K trials",
                "trials_per_update": {"mean": float(np.mean(trials)), "min": float(np.min(trials)), "max": float(np.max(trials)),
                                      "windows_sampled": len(trials)}}

    sharded = None
    if not args.no_sharded and 96 % info.world == 0:
        # every rank reports whether its side is healthy before any collective of this section is entered
        try:
            sharded = time_sharded_window(args, info, dist, backend, dev)
        except Exception as exc:   # noqa: BLE001 -- reported in the JSON line, the headline number stands
            sharded = {"error": f"{type(exc).__name__}: {exc}"}

    if info.rank == 0:
        counts = eng.counts()
        kf_per_s = D.whole_job_throughput(summaries, dt)
        # roofline of the Jacobian-evaluation kernel K1, measured live with HIP events on the
        # engine's own stream: algorithmic bytes of one launch / average launch duration
        k1_ms = eng.time_stage("linearize_imu", reps=20)
        alg_bytes = counts["imu"] * IMU_BYTES
        achieved = alg_bytes / (k1_ms * 1e-3) / 1e9
        stages = {s: eng.time_stage(s, reps=5) for s in
                  ("linearize_imu", "linearize_between", "assemble", "assemble_idle", "solve", "retract", "decide")}
        traffic = measured_traffic_per_imu_factor()
        out = {
            "metric": "keyframes/sec fixed-lag update, 1k-pose window; ATE vs GTSAM ref",
            "value": kf_per_s, "unit": "keyframes/s", "n_gpus": info.world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"fixed-lag update of {args.window}-pose VIL windows (IMU + VIO + LiDAR between "
                                   f"factors at Carla rates), {args.iterations} LM trials per update, "
                                   f"{args.windows} independent windows per GPU",
                       "window_keyframes": args.window, "windows_per_gpu": args.windows,
                       "lm_trials_per_update": args.iterations,
                       "factors_per_gpu": {"imu": counts["imu"], "between": counts["between"]},
                       "parallelism": f"independent windows sharded over {info.world} rank(s), no data-path collective"},
            "roofline": {"kernel": "k_linearize_imu (K1: CombinedImuFactor residual + whitened 15x30 Jacobian)",
                         "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": None if traffic is None else traffic["k1_bytes_per_imu_factor"] * counts["imu"],
                         "traffic_source": None if traffic is None else traffic["source"],
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": k1_ms,
                         "algorithmic_bytes_per_imu_factor": IMU_BYTES,
                         "frac_of_measured_copy_peak_6290": achieved / 6290.0,
                         "priced_on_survey_dense_figure_5496": {
                             "achieved": counts["imu"] * IMU_BYTES_DENSE / (k1_ms * 1e-3) / 1e9,
                             "frac": counts["imu"] * IMU_BYTES_DENSE / (k1_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                             "note": "counts the 159 structurally zero Jacobian entries that are no longer written"}},
            "stage_ms": stages,
            "lm_state_window0": eng.read_lm(0),
        }
        # the kernel that takes most of a step: K4 (banded Cholesky solve), HBM-bound under a full batch.  Algorithmic
        # bytes per keyframe: the band of H it needs (H[k][k-1]: 225, lower triangle of H[k][k]: 120, two 6x6 strips:
        # 432 doubles) + g + the panel written by the forward sweep and read back by the backward one (645 doubles
        # each way) + the increment.
        k4_bytes_per_kf = 8 * (432 + 15 + 645 + 645 + 15)
        n_kf = args.windows * args.window
        k4_ach = n_kf * k4_bytes_per_kf / (stages["solve"] * 1e-3) / 1e9
        out["roofline_solve"] = {"kernel": "k_band_solve (K4: damped block-banded Cholesky factorisation + both substitutions)",
                                 "bound": "hbm", "achieved": k4_ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                 "frac": k4_ach / HBM_PEAK_GBPS, "avg_launch_ms": stages["solve"],
                                 "algorithmic_bytes_per_keyframe": k4_bytes_per_kf,
                                 "algorithmic_bytes_per_launch": n_kf * k4_bytes_per_kf,
                                 "frac_of_measured_copy_peak_6290": k4_ach / 6290.0,
                                 "traffic_note": "profiles/*_pmc_summary.md: k_band_solve read 2*FETCH_SIZE + WRITE_SIZE per launch"}
        if conv is not None:
            out["with_convergence_exit"] = conv
        if sharded is not None:
            out["time_sharded_window"] = sharded
        if not args.no_single_window:
            # latency of the same update on ONE window (what a single vehicle sees)
            one = make_engine(args, 7777, gpu, 1)
            for _ in range(2):
                one_step(one)
            one.sync()
            t1 = time.perf_counter()
            for _ in range(5):
                one_step(one)
            one.sync()
            lat = (time.perf_counter() - t1) / 5
            out["single_window"] = {"ms_per_update": lat * 1e3, "keyframes_per_s": 1.0 / lat,
                                    "solve_ms": one.time_stage("solve", reps=5)}
            one.close()
        if not args.no_degeneracy and info.world == 1:
            out["degeneracy_k6"] = degeneracy_section()
        if not args.no_graph_manager and info.world == 1:
            out["graph_manager"] = graph_manager_section()
        if not args.no_cpu_baseline and info.world == 1:       # the CPU legs: rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(args)
            out["cpu_baseline"]["gpu_over_cpu"] = kf_per_s / out["cpu_baseline"]["value"]
            # also with OpenMP over the factors, on a bounded number of the host's cores
            nthr = max(1, min(16, os.cpu_count() or 1))
            if nthr > 1:
                out["cpu_baseline_openmp"] = cpu_baseline(args, threads=nthr)
        line = json.dumps(out)
    else:
        line = None
    D.barrier(dist)
    if dist is not None:
        dist.destroy_process_group()
    # the ONE JSON line goes out last, after the communication libraries have said whatever they print on stdout
    # (RCCL's version banner, gloo's connection notes)
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)        # RCCL prints its banner through C stdio: push it out before the JSON line
    except OSError:
        pass
    if line is not None:
        print(line, flush=True)


if __name__ == "__main__":
    main()
