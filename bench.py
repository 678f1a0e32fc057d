#!/usr/bin/env python3
"""Headline benchmark: keyframes/sec of fixed-lag smoother updates on 1000-pose windows.

One *step* = one fixed-lag update of every window in the batch: INGEST the new keyframe (vf_engine_ingest_tail: one
host->device copy of its raw IMU samples and between record for all windows, K0 preintegration with each window's current
bias estimate -- GraphManager.cpp:59 -- and staging of the between factor, GraphManager.cpp:83-88), append it (initial
value by IMU prediction, GraphManager.cpp:152-160), marginalise the oldest one into a dense prior (Schur complement,
K-marg), then K Levenberg-Marquardt trials, each = linearise ALL factors of the window (K1+K2) -> block-banded
J^T J (K3) -> banded Cholesky solve (K4) -> retract + cost + accept/reject (K5).
value = (windows on all ranks) * steps / max-over-ranks time: one new keyframe per window per step.

Contract: python bench.py --gpus N --steps K --warmup W ; rank 0 prints ONE JSON line.
N > 1: one rank per GPU -- launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
environment), or, when bench.py is started directly with --gpus N and no WORLD_SIZE, by bench.py itself, which
starts N fresh child ranks BEFORE anything touches the GPU and exits with their status.  Windows are independent,
so ranks share nothing on the data path ("scaling": "weak", windows per GPU fixed).

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
import multiprocessing
import os
import re
import socket
import subprocess
import sys
import threading
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
EXIT_SHARDED_SECTION_HUNG = 75   # the headline line was printed, the time-sharded (collective) section did not finish
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def _make_seq(job):
    from vil_sensor_fusion_amd import synth
    seed, n_kf = job
    return synth.make_sequence(seed=seed, n_kf=n_kf)


def host_workers():
    """Host cores this process can really use: the affinity mask, capped by the cgroup CPU quota (a GPU box shows all
    256 hardware threads in the mask but gives a one-GPU job a share of them)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1") and float(quota) > 0:
                n = min(n, max(1, int(float(quota) / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def make_sequences(args, rank, count, n_kf, base_seed=0):
    """`count` synthetic sequences with distinct seeds (one seed per window), generated on the host
    cores by forked workers -- called BEFORE the process touches the GPU."""
    jobs = [(base_seed + 100003 * rank + s, n_kf) for s in range(count)]
    # (the ranks of one node share its cores: each takes its share, so that N ranks do not fork N x all-cores workers)
    local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))))
    nproc = min(len(jobs), args.host_workers or max(1, host_workers() // local_world), 32)
    if nproc <= 1:
        return [_make_seq(j) for j in jobs]
    with multiprocessing.get_context("fork").Pool(nproc) as pool:
        return pool.map(_make_seq, jobs, chunksize=max(1, len(jobs) // (4 * nproc)))


def updates_per_engine(args):
    """fixed-lag updates the engine must have keyframe slots AND real factors for: warm-up + timed steps, twice when the
    convergence-exit section follows on the same engine"""
    return (args.steps + args.warmup) * (1 if args.no_convergence_exit else 2)


def make_engine(args, local_rank, windows, seqs, updates, all_resident=False, **engine_opts):
    """Synthetic Carla-like factors for `windows` windows.  Only the FIRST window's factors are made resident here (K0 on
    the device, bias estimate 0: GraphManager's bias before the first solve); the factors of the keyframes the updates
    append stay on the host as what a driver receives per keyframe -- raw IMU samples + the between record -- and go in
    through vf_engine_ingest_tail inside the timed step (`feed[u]` = update u's packed arguments for all windows).
    all_resident (timing probes under tools/ only): every factor preintegrated up front, no feed."""
    from vil_sensor_fusion_amd import Engine, EngineOpts, synth
    from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS
    n, total = args.window, args.window + updates + 1
    assert all(s.n >= total for s in seqs), "sequence shorter than the keyframes the run appends"
    eng = Engine(EngineOpts(windows=windows, capacity=total, device=local_rank, **engine_opts))
    nseq = len(seqs)
    recs = [synth.between_records(s) for s in seqs]
    for w in range(windows):
        seq = seqs[w % nseq]
        gt0 = seq.gt_states[0]
        eng.preintegrate(w, 1, seq.imu_off[1:(total if all_resident else n) + 1], seq.imu_steps, np.zeros(6), synth.CARLA_IMU_COV)
        m = seq.btw_b < (total if all_resident else n)
        eng.set_between(w, seq.btw_a[m], seq.btw_b[m], recs[w % nseq][m])
        eng.set_states(w, 0, gt0.reshape(1, 16))
        eng.set_prior(w, 0, synth.prior_record(gt0, REFERENCE_PRIOR_SIGMAS))
        eng.set_range(w, 0, 1)
    eng.predict(-1, 1, n - 1)         # initial values by IMU prediction, all windows in one launch
    for w in range(windows):
        eng.set_range(w, 0, n)
    eng.sync()
    eng.iterate(args.init_iterations)      # converge the initial windows (not timed)
    eng.sync()
    if all_resident:
        return eng, None
    # per sequence and appended keyframe k: its steps and the between factor that ends at k
    by_end = []
    for si, seq in enumerate(seqs):
        d = {}
        for i in np.nonzero(seq.btw_b >= n)[0]:
            d[int(seq.btw_b[i])] = (int(seq.btw_a[i]), recs[si][i])
        by_end.append(d)
    feed, none_rec = [], np.zeros(28)
    for u in range(updates):
        k = n + u
        off, steps, a, rec = [0], [], [], []
        for w in range(windows):
            seq = seqs[w % nseq]
            st = seq.imu_steps[seq.imu_off[k]:seq.imu_off[k + 1]]
            steps.append(st)
            off.append(off[-1] + st.shape[0])
            src, r = by_end[w % nseq].get(k, (-1, none_rec))
            a.append(src)
            rec.append(r)
        feed.append((np.array(off, dtype=np.int32), np.ascontiguousarray(np.concatenate(steps)), synth.CARLA_IMU_COV,
                     np.array(a, dtype=np.int32), np.ascontiguousarray(np.array(rec))))
    return eng, feed


def _cpu_updates(job):
    """`steps` fixed-lag updates of one n-pose window by the CPU oracle, the SAME update the GPU's timed step does
    (tests/helpers.FixedLagOracle: marginalise the leaving keyframe into the dense 27-dof prior and carry it, predict the
    appended keyframe from its IMU factor, K LM trials).  Returns the seconds the updates took (problem construction --
    preintegration, initial values, the initial converging solve -- is outside, as it is outside the GPU's timed region)
    and the window's states after update number `snapshot_at` (what the accuracy figure compares with)."""
    seed, n, n_kf, steps, iterations, threads, snapshot_at, init_iterations = job
    from oracle import oracle
    from tests import helpers
    from vil_sensor_fusion_amd import synth
    seq = synth.make_sequence(seed=seed, n_kf=n_kf)
    assert n_kf >= n + steps + 1
    prob = helpers.build_problem(oracle, seq)
    # (the appended keyframes' factors are preintegrated again at each update with the bias estimate of the moment, as the
    # GPU's ingest does: GraphManager.cpp:59)
    ref = helpers.FixedLagOracle(oracle, prob, n, iterations, threads, init_iterations=init_iterations,
                                 ingest=(seq, oracle.carla_imu_params()))
    snap = ref.window_states.copy() if snapshot_at == 0 else None
    t0 = time.perf_counter()
    for s in range(steps):
        ref.update()
        if s + 1 == snapshot_at:
            snap = ref.window_states.copy()
    return time.perf_counter() - t0, snap, (seq.gt_states[snapshot_at:snapshot_at + n] if snap is not None else None)


def cpu_baseline(args, seq_len):
    """The CPU oracle (a port, not GTSAM: GTSAM cannot be built here) doing the same update on a bounded sample,
    (a) one window on one host core -- the window of seed 0, i.e. GPU window 0 of rank 0, so that its states after
    warmup + steps updates are also the reference of the bench line's `accuracy` -- and (b) one window per host core on ALL
    the cores this process may use (independent windows are how this workload parallelises on a CPU too; the node-level
    figure is the honest one to hold the GPU number against).  Runs before the process touches the GPU (forked workers).
    --no-cpu-baseline keeps only the updates the accuracy check needs and reports no timing."""
    from oracle import oracle
    oracle.build()
    n, cores = args.window, host_workers()
    note = ("C restatement (oracle/vf_oracle.c, gcc -O3 -march=x86-64-v3); the reference's CPU GTSAM path cannot be "
            "built or timed here (no GTSAM/Eigen/Boost/ROS)")
    done = args.steps + args.warmup
    steps1 = done if args.no_cpu_baseline else max(args.cpu_steps, done)
    dt1, snap, gt = _cpu_updates((0, n, seq_len, steps1, args.iterations, 1, done, args.init_iterations))
    reference = dict(states=snap, gt=gt, updates=done, more=[])
    # accuracy is a property of the sequence too (DESIGN.md "Converged start"): a few more windows of the batch (seeds 1, 2,
    # ... = GPU windows 1, 2, ... of rank 0) get a CPU reference as well, one forked worker each
    extra = max(0, min(args.accuracy_windows, args.windows) - 1)
    if extra:
        with multiprocessing.get_context("fork").Pool(min(extra, cores)) as pool:
            res = pool.map(_cpu_updates, [(w, n, seq_len, done, args.iterations, 1, done, args.init_iterations) for w in range(1, extra + 1)], chunksize=1)
        reference["more"] = [dict(window=w + 1, states=r[1], gt=r[2]) for w, r in enumerate(res)]
    if args.no_cpu_baseline:
        return None, None, reference
    one = dict(value=steps1 / dt1, unit="keyframes/s", cores=1, kind="port", host_cores_available=cores,
               sample=f"{steps1} fixed-lag updates of one {n}-pose window (marginalise the oldest keyframe into the dense "
                      f"prior, append + predict one, {args.iterations} LM trials: the update of the GPU's timed step, same "
                      f"sequence as GPU window 0), on 1 core; " + note)
    allc = None
    if cores > 1:
        per = max(4, args.cpu_steps // 4)
        t0 = time.perf_counter()
        with multiprocessing.get_context("fork").Pool(cores) as pool:
            res = pool.map(_cpu_updates, [(1000 + c, n, n + per + 1, per, args.iterations, 1, -1, args.init_iterations) for c in range(cores)], chunksize=1)
        wall = time.perf_counter() - t0
        busy = [r[0] for r in res]
        # rate = updates / the slowest worker's update time (set-up of the problems excluded, as on the GPU side)
        allc = dict(value=cores * per / max(busy), unit="keyframes/s", cores=cores, kind="port",
                    host_cores_available=cores, wall_s_including_setup=wall,
                    sample=f"{cores} independent {n}-pose windows, one per core, {per} fixed-lag updates each (the same "
                           f"marginalised update), {args.iterations} LM trials per update; " + note)
        # SURVEY 8d also names the other way a CPU can use its cores on this workload: ONE window, OpenMP over its factors
        # (the oracle's linearisation loop; assembly, Cholesky and the cost stay serial, as in GTSAM without TBB).  Reported
        # beside the other two: it is the latency form, the one-window-per-core figure above the throughput form.
        try:
            dt_omp, _, _ = _cpu_updates((2000, n, n + per + 1, per, args.iterations, cores, -1, args.init_iterations))
            allc["one_window_openmp_over_factors"] = {"value": per / dt_omp, "unit": "keyframes/s", "threads": cores,
                                                      "sample": f"{per} updates of one {n}-pose window, linearisation of its factors "
                                                                f"on {cores} OpenMP threads, the rest serial"}
        except Exception as exc:   # noqa: BLE001
            allc["one_window_openmp_over_factors"] = {"error": f"{type(exc).__name__}: {exc}"}
    return one, allc, reference


def accuracy_vs_oracle(gpu_states, reference):
    """ATE = sqrt(mean |t_est - t_ref|^2) over the window's keyframes, no alignment (gauge fixed by the priors),
    and the largest rotation error 2 acos|q_w| of q_est^-1 q_ref (gtsam_fusion/python/diagnostics.py:114,122) -- GPU window 0
    after the timed region against the CPU oracle after the same updates of the same sequence; and both against the
    synthetic ground truth."""
    from tests import helpers
    ate, rot = helpers.ate(gpu_states, reference["states"])
    gt_gpu, _ = helpers.ate(gpu_states, reference["gt"])
    gt_cpu, _ = helpers.ate(reference["states"], reference["gt"])
    return {"ate_m": ate, "rot_rad": rot, "updates": reference["updates"], "window": 0, "keyframes": int(gpu_states.shape[0]),
            "vs": "CPU oracle (oracle/vf_oracle.c) after the same marginalised fixed-lag updates of the same sequence; "
                  "GTSAM itself cannot be run here", "bar_m": 1e-6, "within_bar": bool(ate <= 1e-6 and rot <= 1e-6),
            "ate_vs_ground_truth_m": {"gpu": gt_gpu, "cpu_oracle": gt_cpu}}


def accuracy_vs_independent_qr(gpu_states, reference, args, seq_len, done):
    """GPU window 0 (and the CPU oracle) against the committed trajectory of the INDEPENDENT optimiser (oracle/twin_qr.py: the
    twin's own preintegration and residuals, automatic-differentiation Jacobians, Householder QR on the whitened Jacobian
    as the reference's iSAM2 is configured to factorise -- GraphManager.cpp:38 --, no normal equations, its own accept
    rule; converged at every update).  The fixture (tests/golden/qr_twin_fixed_lag.npz, made by
    tests/golden/make_qr_twin_golden.py) is data; it exists for this window length / sequence length at a few update counts."""
    from tests import helpers
    path = os.path.join(ROOT, "tests", "golden", "qr_twin_fixed_lag.npz")
    if not os.path.exists(path):
        return {"skipped": "fixture missing"}
    F = np.load(path)
    if int(F["window"]) != args.window or int(F["seq_len"]) != seq_len or done not in F["updates"]:
        return {"skipped": f"the fixture holds window {int(F['window'])}, sequence length {int(F['seq_len'])}, updates "
                           f"{[int(u) for u in F['updates']]}; this run: {args.window}, {seq_len}, {done}"}
    ref = np.zeros((args.window, 16))
    ref[:, :7] = F["states_u0"][:, :7] if done == 0 else F[f"pose_u{done}"]
    ate, rot = helpers.ate(gpu_states, ref)
    out = {"ate_m": ate, "rot_rad": rot, "updates": done, "window": 0, "bar_m": 1e-6, "within_bar": bool(ate <= 1e-6 and rot <= 1e-6),
           "vs": "oracle/twin_qr.py: independent residuals + AD Jacobians + Householder QR elimination (no normal equations), "
                 "converged at every update; fixture tests/golden/qr_twin_fixed_lag.npz",
           "twin_last_gauss_newton_step": float(F["last_polish_step_per_update"][done - 1]) if done > 0 else float(F["polish_steps_u0"][-1])}
    if reference is not None and reference.get("states") is not None:
        out["cpu_oracle_vs_independent_qr_ate_m"] = helpers.ate(reference["states"], ref)[0]
    return out


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
    # A window this long is past what float64 normal equations resolve (cond ~ n^4): the library refines every solve by
    # conjugate gradients through J -- R more solves with the same factor shape, two collectives each -- and accepts trials
    # non-monotonically (DESIGN.md "Refined solve").  First converge from dead reckoning (12 trials, not timed) ...
    refine = eng.refine_count()
    solver.iterate(12)
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
           "refine_corrections_per_solve": refine, "ms_per_solve": per_trial * 1e3 / (1 + refine),
           "keyframe_relinearisations_per_s": n / per_trial,
           "exchange_doubles_per_solve": chunks * 2248 + (n + 8 + 63) // 64 * 64 * 15 + 1,   # separator slots; increments of every slot + failure flag
           "collectives_per_trial": D.ShardedSolver.COLLECTIVES_PER_TRIAL * (1 + refine),
           "collectives": "per solve 1 in-place all-gather (packed separator system) + 1 all-reduce (increments + failure flags), "
                          "1 + refine_corrections_per_solve solves per trial; the cost and the operator J^T J need none (every rank "
                          "holds every residual and every Jacobian)",
           "collectives_issued": solver.collectives, "lm_trials_run": 12 + reps * trials,
           "backend": backend if dist is not None else "none", "final_cost": lm["cost"], "solve_failures": lm["solve_failures"],
           "lm": {"accepted": lm["accepted"], "rejected": lm["rejected"], "provisional": eng.read_excursions(0)[0]}}
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "qr_twin_10k.npz")
    if n == 10000 and os.path.exists(gold):
        # ... and say where it ended: against the optimum an independent Householder-QR optimiser found for this window
        F = np.load(gold)
        x = eng.get_states(0, 0, n)
        d = x[:, 4:7] - F["states"][:, 4:7]
        out["vs_independent_qr_optimum"] = {"ate_m": float(np.sqrt(np.mean(np.sum(d * d, axis=1)))), "qr_final_cost": float(F["final_cost"]),
                                            "started_from": "IMU dead reckoning, 15 m (ATE) from it"}
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
    # roofline of K6 on a launch big enough to be bandwidth- rather than launch-bound (>= 2^20 units per launch;
    # here 2^22 matrices = 1.2 GB of float64 input): algorithmic bytes = 288 B read + 8 B written per 6x6 (144 + 4 in float32)
    T2 = 1 << 22
    big = np.ascontiguousarray(np.tile(mats[:, :, :1 << 16], (1, 1, T2 >> 16)))
    roof = {"matrices": T2, "bound": "hbm", "peak": HBM_PEAK_GBPS, "unit": "GB/s", "kernels": {}}
    for name in ("d_opt", "e_opt", "condition_number"):
        for dt, tag, nbytes in ((np.float64, "f64", 296), (np.float32, "f32", 148)):
            _, ms = dg.apply_degen_function(big, None, "all", name, dtype=dt, reps=5)
            ach = T2 * nbytes / (ms * 1e-3) / 1e9
            roof["kernels"][f"{name}/{tag}"] = {"avg_launch_ms": ms, "achieved": ach, "frac": ach / HBM_PEAK_GBPS,
                                                "algorithmic_bytes_per_matrix": nbytes, "ns_per_matrix": ms * 1e6 / T2}
    for dt, tag, nbytes in ((np.float64, "f64", 296 + 16), (np.float32, "f32", 148 + 8)):
        _, ms = dg.spectrum(big, "all", dtype=dt, reps=5)
        ach = T2 * nbytes / (ms * 1e-3) / 1e9
        roof["kernels"][f"spectrum(e_opt+max_eigen+condition_number)/{tag}"] = {"avg_launch_ms": ms, "achieved": ach, "frac": ach / HBM_PEAK_GBPS,
                                                                             "algorithmic_bytes_per_matrix": nbytes, "ns_per_matrix": ms * 1e6 / T2}
    roof["note"] = ("d_opt (pivoted LU in registers) is the metric the shipped gate uses and is HBM-bound.  e_opt / max_eigen / condition_number "
                    "(round 6: Householder tridiagonalisation + implicit QL with deflation in registers, the ends of the spectrum of ONE "
                    "eigen-solve; condition_number of a symmetric matrix from it instead of a Jacobi SVD) are bound by float64 VALU issue, "
                    "not HBM: their HBM fraction is quoted for completeness (round 5, cyclic Jacobi: 0.23 / 0.56 ns per matrix)")
    out["roofline_k6"] = roof
    return out


def graph_manager_section(lag=1000, extra=120):
    """Latency of the drop-in surface itself: GraphManager.solve() (= vf_solve) per keyframe in fixed-lag mode, fed like
    the node (IMU at 200 Hz between keyframes, one reserveNode per camera / LiDAR keyframe, the VIO and LiDAR between
    factors of the synthetic sequence, one solve per keyframe), once the window is full.  Two settings: the default
    (LM trials stop by GTSAM's rule, vf_graph_opts.rel_tol / abs_tol = 1e-5) and always 5 trials."""
    from vil_sensor_fusion_amd import GraphManager, synth
    nkf = lag + extra
    seq = synth.make_sequence(seed=3, n_kf=nkf + 2)
    out = {"lag_keyframes": lag, "lm_trials_per_solve_max": 5,
           "what": "vf_solve: K0 of the new IMU factor, prediction, staging of the new between factor, marginalisation of "
                   "the oldest keyframe, LM trials (warm start: only the appended tail is linearised first), read-back"}
    # paced: the device is idle when vf_solve is called, as at the node's 20-30 Hz keyframe rate (what the library enqueues behind a
    # solve -- K0 of the next factor at reserveNode, the next marginal prior -- has run); the other two call it back to back
    for name, tol, paced in (("default_termination", None, False), ("default_termination_paced", None, True), ("always_5_trials", 0.0, False)):
        gm = GraphManager(capacity=lag + 192, lag=lag, iterations=5, rel_tol=tol, abs_tol=tol)
        gm.setInitialState(seq.gt_states[0])      # the synthetic vehicle is already moving at t = 0 (the reference's anchor is "at rest")
        gm.addIMUMeasurement(0.0, seq.imu_steps[0, 1:4], seq.imu_steps[0, 4:7])
        t, times = 0.0, []
        for k in range(1, nkf):
            for s in seq.imu_steps[seq.imu_off[k]:seq.imu_off[k + 1]]:
                t += s[0]
                gm.addIMUMeasurement(t, s[1:4], s[4:7])
            gm.reserveNode(t)
            for i in np.nonzero(seq.btw_b == k)[0]:
                if seq.btw_a[i] >= 0:
                    gm.addBetweenFactor(int(seq.btw_a[i]), k, (seq.btw_q[i], seq.btw_t[i]), np.eye(6) * seq.btw_cov[i])
            if paced and k > lag:
                gm.lmStats()          # (synchronises the engine's stream)
            t0 = time.perf_counter()
            gm.solve()
            times.append(time.perf_counter() - t0)
        steady = np.array(times[lag + 20:]) * 1e3
        gm.close()
        out[name] = {"solve_ms_mean": float(steady.mean()), "solve_ms_median": float(np.median(steady)), "solve_ms_p99": float(np.percentile(steady, 99)),
                     "solves_timed": int(steady.size)}
    out["solve_ms_mean"] = out["default_termination"]["solve_ms_mean"]
    out["paced_solve_ms_mean"] = out["default_termination_paced"]["solve_ms_mean"]
    out["paced_solve_ms_median"] = out["default_termination_paced"]["solve_ms_median"]
    out["paced_solve_ms_p99"] = out["default_termination_paced"]["solve_ms_p99"]
    # ... and with a loop closure alive: a far factor (GraphManager.cpp:83-88 takes any pair of keys) costs six more band
    # solves per LM trial while it is in the window -- nonlinear at first, then, once its older key has been marginalised,
    # as the engine's linear far factor
    try:
        gm = GraphManager(capacity=lag + 192, lag=lag, iterations=5)
        gm.setInitialState(seq.gt_states[0])
        gm.addIMUMeasurement(0.0, seq.imu_steps[0, 1:4], seq.imu_steps[0, 4:7])
        t, times, a0, b0 = 0.0, {}, 40, lag - 100
        for k in range(1, nkf):
            for st in seq.imu_steps[seq.imu_off[k]:seq.imu_off[k + 1]]:
                t += st[0]
                gm.addIMUMeasurement(t, st[1:4], st[4:7])
            gm.reserveNode(t)
            for i in np.nonzero(seq.btw_b == k)[0]:
                if seq.btw_a[i] >= 0:
                    gm.addBetweenFactor(int(seq.btw_a[i]), k, (seq.btw_q[i], seq.btw_t[i]), np.eye(6) * seq.btw_cov[i])
            if k == b0:                                  # the closure a0 <-> b0, measured as the ground truth has it
                Ra, Rb = synth.quat_to_rot(seq.gt_states[a0, :4]), synth.quat_to_rot(seq.gt_states[b0, :4])
                gm.addBetweenFactor(a0, b0, (synth.rot_to_quat(Ra.T @ Rb), Ra.T @ (seq.gt_states[b0, 4:7] - seq.gt_states[a0, 4:7])), np.eye(6) * 1e-4)
            t0 = time.perf_counter()
            gm.solve()
            times[k] = (time.perf_counter() - t0) * 1e3
        st = gm.lmStats()
        gm.close()
        nonlin = [times[k] for k in range(b0 + 5, lag + a0 - 2)]            # both keys in the window
        linear = [times[k] for k in range(lag + a0 + 8, nkf)]               # the older key marginalised, the closure a linear far factor
        out["with_a_loop_closure"] = {"keys": [a0, b0], "solve_ms_mean_nonlinear_far_factor": float(np.mean(nonlin)) if nonlin else None,
                                      "solve_ms_mean_linear_far_factor": float(np.mean(linear)) if linear else None,
                                      "solves_timed": [len(nonlin), len(linear)], "solve_failures": st["solve_failures"]}
    except Exception as exc:   # noqa: BLE001
        out["with_a_loop_closure"] = {"error": f"{type(exc).__name__}: {exc}"}
    return out


def incremental_section(args, gpu, seqs):
    """`incremental_update`: the fixed-lag update as ONE Gauss-Newton update about per-keyframe linearisation points (what
    GraphManager::solve does per keyframe: ISAM2::update with relinearizeThreshold 1e-4, GraphManager.cpp:37-43,126-127), done
    incrementally: ingest + marginalising slide + vf_engine_isam_step on an engine made with incremental = 1."""
    from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS
    sv = argparse.Namespace(**vars(args))
    n_up = args.steps + args.warmup
    eng, feed = make_engine(sv, gpu, args.windows, seqs, n_up, incremental=1, chunks=1, sweep_two_sided_max=0, solve_assemble_min=0,
                            refine_iterations=0, lm_excursion=0)
    try:
        eng.isam_step(1e-4)                  # the first update of an incremental engine covers the whole window
        u = 0

        def step():
            nonlocal u
            eng.ingest_tail(*feed[u])
            u += 1
            eng.slide(REFERENCE_PRIOR_SIGMAS, marginalize=True)
            eng.isam_step(1e-4)
        for _ in range(args.warmup):
            step()
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        eng.sync()
        dt = time.perf_counter() - t0
        sample = range(0, args.windows, max(1, args.windows // 64) | 1)
        infos = [eng.incremental_info(w) for w in sample]
        lo = n_up
        frac = [(lo + args.window - i["first_eliminated"]) / args.window for i in infos]
        sub = [(lo + args.window - i["last_substituted"]) / args.window for i in infos]
        est = eng.get_estimate(0, n_up, args.window)
        d = est[:, 4:7] - seqs[0].gt_states[n_up:n_up + args.window, 4:7]
        return {"ms_per_step_this_rank": dt / args.steps * 1e3,
                "what": "per update and window: ingest (K0 at the current bias), marginalising slide, ONE Gauss-Newton update with relinearisation "
                        "threshold 1e-4 done incrementally (vf_engine_isam_step, vf_engine_opts.incremental); same windows and sequences as the headline",
                "relinearised_frac": float(np.mean(frac)),
                "window_fraction_eliminated_again": {"mean": float(np.mean(frac)), "min": float(np.min(frac)), "max": float(np.max(frac)), "windows_sampled": len(frac)},
                "window_fraction_substituted_again": {"mean": float(np.mean(sub)), "min": float(np.min(sub))},
                "whole_window_updates": infos[0]["whole_window_updates"], "updates": infos[0]["updates"],
                "ate_vs_ground_truth_m_window0": float(np.sqrt(np.mean(np.sum(d * d, axis=1)))),
                "note": "on this stream every update moves the whole window by more than the reference's threshold (stiff IMU chain, soft odometry): "
                        "the suffix that is eliminated again is the window, as it would be for iSAM2; the rate is that of one update per keyframe "
                        "against the headline's K LM trials"}
    finally:
        eng.close()


def measured_traffic_per_imu_factor():
    """HBM bytes per IMU factor of K1 from the committed PMC profile (separate --pmc passes,
    2*FETCH_SIZE + WRITE_SIZE, KiB units; tools/summarize_prof.py)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return None
    tr = json.load(open(path))
    # the PMC passes and the kernel trace must come from ONE profiling round (tools/profile_round.sh writes both): a
    # traffic file left over from an older round than kernel_durations.json is refused, not silently quoted
    prof = profiled_kernels()
    tag = lambda d: (re.search(r"profiles/(r\d+\w*?)_", d.get("source", "")) or [None, None])[1]
    if prof is not None and tag(prof) != tag(tr):
        return {"stale": f"profiles/traffic.json is of round-tag {tag(tr)}, kernel_durations.json of {tag(prof)}: not used"}
    return tr


def spawn_ranks(args, script=None, argv=None):
    """python bench.py --gpus N with no WORLD_SIZE in the environment: start N fresh ranks of this script (one per GPU,
    RCCL by default) and exit with their status.  Nothing in this parent process has touched the GPU (only argparse,
    numpy and the standard library are loaded), and the parent never re-executes itself."""
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] +
                                      (sys.argv[1:] if argv is None else list(argv)), env=env))
    rc = 0
    for pr in procs:
        code = pr.wait()
        rc = rc or code
    return rc


def profiled_kernels():
    """Per-kernel durations of the committed rocprofv3 trace of this bench (profiles/kernel_durations.json, written
    by tools/summarize_prof.py) -- carried in the line beside the live HIP-event figures so that every quoted
    fraction can be recomputed from one committed file."""
    path = os.path.join(ROOT, "profiles", "kernel_durations.json")
    return json.load(open(path)) if os.path.exists(path) else None


def roofline_k1(eng):
    """roofline of the Jacobian-evaluation kernel K1, measured live with HIP events on the engine's own stream:
    algorithmic bytes of one launch / average launch duration; PMC traffic from the committed counter passes."""
    counts = eng.counts()
    k1_ms = eng.time_stage("linearize_imu", reps=20)
    alg_bytes = counts["imu"] * IMU_BYTES
    achieved = alg_bytes / (k1_ms * 1e-3) / 1e9
    traffic = measured_traffic_per_imu_factor()
    return {"kernel": "k_linearize_imu (K1: CombinedImuFactor residual + whitened 15x30 Jacobian)",
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": None if (traffic is None or "stale" in traffic) else traffic["k1_bytes_per_imu_factor"] * counts["imu"],
            "traffic_source": None if traffic is None else traffic.get("source", traffic.get("stale")),
            "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": k1_ms,
            "avg_launch_ms_source": "HIP events on the engine's stream, 20 back-to-back launches (vf_engine_time_stage)",
            "algorithmic_bytes_per_imu_factor": IMU_BYTES,
            "frac_of_measured_copy_peak_6290": achieved / 6290.0,
            "priced_on_survey_dense_figure_5496": {
                "achieved": counts["imu"] * IMU_BYTES_DENSE / (k1_ms * 1e-3) / 1e9,
                "frac": counts["imu"] * IMU_BYTES_DENSE / (k1_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "note": "counts the 159 structurally zero Jacobian entries that are no longer written"}}


LINE_LIMIT = 4096        # the driver keeps the tail of stdout: the ONE JSON line must fit well inside it


def _r(x, sig=6):
    """floats at `sig` significant digits (the line is a summary; bench_detail.json keeps full precision)"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        return float(f"{x:.{sig}g}") if np.isfinite(x) else None
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return _r(float(x), sig)


def _pick(d, *keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def compact_line(full, detail_name="bench_detail.json"):
    """The ONE JSON line the driver parses: the contract keys, `roofline`, `cpu_baseline` and a handful of labelled scalars,
    <= LINE_LIMIT characters.  Everything else bench.py measures (per-kernel profile, issue view, K6 table, the
    time-sharded window, prose) is in `full`, which main() writes to bench_detail.json beside this script."""
    out = _pick(full, "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data")
    out["vs_baseline"] = full.get("vs_baseline")
    cfg = full.get("config", {})
    out["config"] = _pick(cfg, "workload", "window_keyframes", "windows_per_gpu", "lm_trials_per_update", "factors_per_gpu", "parallelism")
    acc = full.get("accuracy")
    if acc:
        out["accuracy"] = _pick(acc, "ate_m", "rot_rad", "ate_m_max", "within_bar_all", "updates", "bar_m")
        out["accuracy"]["vs"] = "CPU oracle, same updates (GTSAM cannot be run here)"
        qr = acc.get("vs_independent_qr") or {}
        if "ate_m" in qr:
            out["accuracy"]["vs_independent_qr_ate_m"] = qr["ate_m"]
    if "stage_ms" in full:
        out["stage_ms"] = _r({k: v for k, v in full["stage_ms"].items() if not k.startswith("assemble_k3")}, 4)
    if "solve_form" in full:
        out["solve_form"] = full["solve_form"]
    rf = full.get("roofline")
    if rf:
        out["roofline"] = _pick(rf, "kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic",
                                "avg_launch_ms", "profiled_avg_launch_ms", "profiled_frac", "profiled_source",
                                "algorithmic_bytes_per_imu_factor", "frac_of_measured_copy_peak_6290")
        out["roofline"].setdefault("traffic", None)
        out["roofline"]["kernel"] = "k_linearize_imu (K1)"
    rs = full.get("roofline_solve")
    if rs:
        out["roofline_solve"] = _pick(rs, "bound", "achieved", "peak", "unit", "frac", "avg_launch_ms")
        out["roofline_solve"]["kernel"] = "band solve (K3+K4)" if full.get("solve_form") == "assembling" else "k_band_solve (K4)"
        mv = rs.get("matrix_instruction_view")
        if mv:
            out["roofline_solve"]["mfma_f64_tflops"] = mv["achieved"]
    cb = full.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = _pick(cb, "value", "unit", "cores", "kind")
        out["cpu_baseline"]["sample"] = (f"{cfg.get('window_keyframes')}-pose window, same fixed-lag update, "
                                         f"{cfg.get('lm_trials_per_update')} LM trials, C oracle on 1 core (not GTSAM: cannot be built here)")
    ca = full.get("cpu_baseline_all_cores")
    if ca:
        out["cpu_baseline_all_cores"] = _pick(ca, "value", "cores")
    for key, sub in (("with_convergence_exit", ("value", "ms_per_step", "error")),
                     ("incremental_update", ("value", "ms_per_step", "relinearised_frac", "error")),
                     ("single_window", ("ms_per_update", "solve_ms")),
                     ("graph_manager", ("solve_ms_mean", "paced_solve_ms_mean", "paced_solve_ms_median", "paced_solve_ms_p99"))):
        if isinstance(full.get(key), dict):
            out[key] = _pick(full[key], *sub)
    sw = full.get("single_window")
    if isinstance(sw, dict) and cb and sw.get("ms_per_update") and cb.get("value"):
        # the like-for-like figure for ONE vehicle: one window on the GPU against one window on one CPU core (the headline holds
        # 1 024 windows against one; a GPU-over-CPU ratio is not a measure of kernel quality either way)
        out["single_window"]["over_cpu_one_core_one_window"] = (1e3 / sw["ms_per_update"]) / cb["value"]
    gm = full.get("graph_manager")
    if isinstance(gm, dict) and isinstance(gm.get("default_termination"), dict):
        out["graph_manager"]["solve_ms_p99"] = gm["default_termination"].get("solve_ms_p99")
    ts = full.get("time_sharded_window")
    if isinstance(ts, dict):
        out["time_sharded_window"] = _pick(ts, "window_keyframes", "ranks", "ms_per_lm_trial", "error")
    k6 = (full.get("degeneracy_k6") or {}).get("roofline_k6", {}).get("kernels")
    if k6:
        out["degeneracy_k6_ns_per_matrix"] = {k: v["ns_per_matrix"] for k, v in k6.items() if k.endswith("/f64")}
    out["detail"] = detail_name
    out = _r(out)
    line = json.dumps(out, separators=(",", ":"))
    # never print a line the driver cannot keep: shed the optional objects, least important first
    for victim in ("degeneracy_k6_ns_per_matrix", "time_sharded_window", "stage_ms", "roofline_solve", "single_window",
                   "graph_manager", "accuracy", "cpu_baseline_all_cores", "with_convergence_exit", "incremental_update"):
        if len(line) <= LINE_LIMIT:
            break
        out.pop(victim, None)
        line = json.dumps(out, separators=(",", ":"))
    assert len(line) <= LINE_LIMIT, len(line)
    return line


def write_detail(full, path=None):
    path = path or os.path.join(os.environ.get("VF_BENCH_DETAIL_DIR", ROOT), "bench_detail.json")
    try:
        with open(path, "w") as f:
            json.dump(full, f, indent=1)
        return path
    except OSError as exc:       # a read-only checkout must not cost the line
        print(f"bench.py: could not write {path}: {exc}", file=sys.stderr)
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--window", type=int, default=1000, help="keyframes per window (BASELINE metric: 1000)")
    ap.add_argument("--windows", type=int, default=1024, help="independent windows per GPU")
    ap.add_argument("--sequences", type=int, default=0,
                    help="distinct synthetic sequences (seeds) per rank; 0 = one per window (SURVEY 8d)")
    ap.add_argument("--iterations", type=int, default=5, help="LM trials per update")
    ap.add_argument("--init-iterations", type=int, default=200,
                    help="LM trials of the untimed solve that converges every window before the first update (GPU and CPU "
                         "legs alike).  A 1000-pose window started from IMU dead reckoning needs 50-150 to converge; slid "
                         "while it is still far from its optimum, the soft global yaw / position mode of the marginal prior "
                         "keeps two float64 implementations 1e-5 m apart (DESIGN.md, Converged start)")
    ap.add_argument("--cpu-steps", type=int, default=64,
                    help="fixed-lag updates the one-core CPU baseline is timed on (64 = about 5 s)")
    ap.add_argument("--host-workers", type=int, default=0,
                    help="processes that generate the synthetic sequences (0 = the host cores this job may use; 1 = no worker "
                         "processes, e.g. under a profiler)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--accuracy-windows", type=int, default=4,
                    help="windows of rank 0 (0 .. n-1) whose final states are compared with the CPU oracle doing the same updates")
    ap.add_argument("--no-accuracy", action="store_true", help="skip the CPU oracle altogether (no accuracy object, no cpu_baseline)")
    ap.add_argument("--no-single-window", action="store_true")
    ap.add_argument("--reanchor", action="store_true", help="drop the oldest keyframe by re-anchoring tight priors instead of marginalising it")
    ap.add_argument("--sharded-window", type=int, default=10000, help="keyframes of the time-sharded window (BASELINE configs[4])")
    ap.add_argument("--no-sharded", action="store_true")
    ap.add_argument("--sharded-timeout", type=float, default=240.0,
                    help="seconds after which a multi-rank run abandons the time-sharded section and prints the headline")
    ap.add_argument("--no-convergence-exit", action="store_true")
    ap.add_argument("--no-incremental", action="store_true")
    ap.add_argument("--no-degeneracy", action="store_true")
    ap.add_argument("--no-graph-manager", action="store_true")
    ap.add_argument("--solve-assemble-min", type=int, default=None,
                    help="vf_engine_opts.solve_assemble_min of the headline engine (default: the library's)")
    ap.add_argument("--solve-assemble-waves", type=int, default=None,
                    help="vf_engine_opts.solve_assemble_waves of the headline engine (default: the library's)")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_env}: launch with "
                 f"`python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus}` "
                 f"or start `python bench.py --gpus {args.gpus}` without WORLD_SIZE set")
    rank_env = int(os.environ.get("RANK", "0"))

    # ---- host-only phase: everything that forks worker processes happens before the first GPU call
    cpu_one = cpu_all = reference = None
    updates = updates_per_engine(args)
    # every sequence is generated at ONE length (make_sequence is not prefix-stable in its length): what the engine's
    # updates need, and what the one-core CPU leg needs to run its sample on the sequence of GPU window 0
    if world_env > 1:
        args.no_cpu_baseline = True      # the CPU legs are timed at N = 1 only; at N > 1 just the updates the accuracy figure needs
    seq_len = args.window + max(updates, 0 if (args.no_cpu_baseline or args.no_accuracy) else args.cpu_steps) + 1
    if rank_env == 0 and not args.no_accuracy:
        cpu_one, cpu_all, reference = cpu_baseline(args, seq_len)
    nseq = args.sequences if args.sequences > 0 else args.windows
    seqs = make_sequences(args, rank_env, max(1, min(nseq, args.windows)), seq_len)
    one_updates = 7
    one_seq = make_sequences(args, rank_env, 1, args.window + one_updates + 1, base_seed=7777)

    import torch
    from vil_sensor_fusion_amd import distributed as D
    from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS
    info = D.rank_info()
    dist = None
    # VF_BENCH_BACKEND=gloo + VF_BENCH_SHARE_GPU=1 let the ranks share GPU 0 (control-plane smoke test
    # on a 1-GPU box); the driver's N-GPU runs use RCCL ("nccl") with one rank per GPU.
    backend = os.environ.get("VF_BENCH_BACKEND", "nccl")
    gpu = 0 if os.environ.get("VF_BENCH_SHARE_GPU") else info.local_rank
    if info.world > 1 or os.environ.get("VF_FORCE_DIST"):
        torch.cuda.set_device(gpu)
        dist = D.init(backend=backend, device_id=torch.device("cuda", gpu) if backend == "nccl" else None)
    dev = torch.device("cuda", gpu)

    eng, feed = make_engine(args, gpu, args.windows, seqs, updates,
                            **({} if args.solve_assemble_min is None else {"solve_assemble_min": args.solve_assemble_min}),
                            **({} if args.solve_assemble_waves is None else {"solve_assemble_waves": args.solve_assemble_waves}))
    fed = {id(eng): 0}

    def fence():
        D.barrier(dist)
        torch.cuda.synchronize(dev)
        eng.sync()

    def one_step(e, feed_=None):
        f = feed if feed_ is None else feed_
        u = fed.setdefault(id(e), 0)
        fed[id(e)] = u + 1
        e.ingest_tail(*f[u])               # H2D of the new keyframe's samples + between record, K0 with the current bias, staging
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
    ingest_ms = eng.ingest_status()        # raises if any ingest of the run failed on the device; (h2d, K0 + staging) of the last one
    lm_after_timed = eng.read_lm(0)
    accuracy = None
    if reference is not None and reference["states"] is not None:
        done = args.steps + args.warmup
        accuracy = accuracy_vs_oracle(eng.get_states(0, done, args.window), reference)
        per = [dict(window=0, ate_m=accuracy["ate_m"], rot_rad=accuracy["rot_rad"])]
        for m in reference.get("more", []):
            a = accuracy_vs_oracle(eng.get_states(m["window"], done, args.window), dict(states=m["states"], gt=m["gt"], updates=done))
            per.append(dict(window=m["window"], ate_m=a["ate_m"], rot_rad=a["rot_rad"]))
        accuracy["windows"] = per
        accuracy["vs_independent_qr"] = accuracy_vs_independent_qr(eng.get_states(0, done, args.window), reference, args, seq_len, done)
        accuracy["ate_m_max"] = max(x["ate_m"] for x in per)
        accuracy["ate_m_median"] = float(np.median([x["ate_m"] for x in per]))
        accuracy["within_bar_all"] = bool(all(x["ate_m"] <= 1e-6 and x["rot_rad"] <= 1e-6 for x in per))
        accuracy["note"] = ("ate_m / rot_rad are window 0; `windows` lists every compared window.  Two float64 normal-equation "
                            "solvers agree on a 1000-pose fixed-lag window to (cond * eps) x (how far the window's soft modes move "
                            "per update): 1e-8 ... 1e-6 m depending on the sequence (DESIGN.md, Converged start; "
                            "tools/accuracy_sweep.py: 16 sequences, median 5e-8, worst 1.4e-6)")

    # Same update with GTSAM's LM termination rule switched on (vf_engine_set_convergence): a second,
    # clearly labelled number -- the headline above always runs all K trials on every window.  It continues on the
    # same engine (whose capacity and sequences cover these updates too); a failure here is reported in the line and
    # never costs the headline.  No collective sits inside the try block.
    conv = None
    if not args.no_convergence_exit:
        try:
            sample = range(0, args.windows, max(1, args.windows // 64) | 1)   # ~64 windows, odd stride
            eng.set_convergence(1e-5, 1e-5)
            for _ in range(args.warmup):
                one_step(eng)
            eng.sync()
            before = [eng.read_lm(w) for w in sample]
            t1 = time.perf_counter()
            for _ in range(args.steps):
                one_step(eng)
            eng.sync()
            dt_c = time.perf_counter() - t1
            after = [eng.read_lm(w) for w in sample]
            trials = [(a["accepted"] + a["rejected"] - b["accepted"] - b["rejected"]) / args.steps for a, b in zip(after, before)]
            eng.set_convergence(0.0, 0.0)
            conv = {"ms_per_step_this_rank": dt_c / args.steps * 1e3,
                    "rule": "a window stops after a trial that changes its cost by <= 1e-5 absolute or relative "
                            "(gtsam LevenbergMarquardtParams defaults), at most K trials",
                    "trials_per_update": {"mean": float(np.mean(trials)), "min": float(np.min(trials)), "max": float(np.max(trials)),
                                          "windows_sampled": len(trials), "updates_counted": args.steps}}
        except Exception as exc:   # noqa: BLE001
            conv = {"error": f"{type(exc).__name__}: {exc}"}
        slow = D.max_over_ranks(dist, conv.get("ms_per_step_this_rank", float("inf")),
                                device=dev if (dist is not None and backend == "nccl") else "cpu")
        if "error" not in conv and np.isfinite(slow):
            conv["ms_per_step"] = slow
            conv["value"] = info.world * args.windows / (slow * 1e-3)
            conv["unit"] = "keyframes/s"

    # The update done the way the reference does it -- ONE iSAM2-like Gauss-Newton update per keyframe instead of K LM trials --
    # and incrementally (vf_engine_opts.incremental: only the keyframes from the first one that moved are linearised, assembled and
    # eliminated again; DESIGN.md "Incremental updates").  A third, clearly labelled number, on an engine of its own.
    incr = None
    if not args.no_incremental:
        try:
            incr = incremental_section(args, gpu, seqs)
        except Exception as exc:   # noqa: BLE001
            incr = {"error": f"{type(exc).__name__}: {exc}"}
        slow = D.max_over_ranks(dist, incr.get("ms_per_step_this_rank", float("inf")), device=dev if (dist is not None and backend == "nccl") else "cpu")
        if "error" not in incr and np.isfinite(slow):
            incr["ms_per_step"] = slow
            incr["value"] = info.world * args.windows / (slow * 1e-3)
            incr["unit"] = "keyframes/s"
    del seqs

    # K1's roofline is measured before the one section that holds collectives, so that the fallback line of a stalled
    # multi-rank run carries it too
    roofline = roofline_k1(eng) if info.rank == 0 else None

    sharded = None
    if not args.no_sharded and 96 % info.world == 0:
        # every rank reports whether its side is healthy before any collective of this section is entered
        guard = None
        if info.world > 1:
            # The only part of the run whose collectives sit on the data path.  It has never run on more than one GPU's
            # worth of hardware here; should it ever stall in a collective, the headline measured above must still
            # reach the driver: after `--sharded-timeout` seconds every rank leaves, rank 0 with the line it has.
            def bail():
                if info.rank == 0:
                    kfs = D.whole_job_throughput(summaries, dt)
                    fb = {
                        "metric": "keyframes/sec fixed-lag update, 1k-pose window; ATE vs GTSAM ref", "value": kfs,
                        "unit": "keyframes/s", "n_gpus": info.world, "steps": args.steps, "warmup": args.warmup,
                        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                        "dtype": "f64", "data": "synthetic",
                        "config": {"workload": f"fixed-lag update of {args.window}-pose VIL windows, {args.iterations} LM trials "
                                               f"per update, {args.windows} independent windows per GPU",
                                   "window_keyframes": args.window, "windows_per_gpu": args.windows, "lm_trials_per_update": args.iterations,
                                   "parallelism": f"independent windows sharded over {info.world} rank(s), no data-path collective"},
                        "accuracy": accuracy, "roofline": roofline,
                        "with_convergence_exit": conv, "incremental_update": incr,
                        "time_sharded_window": {"error": f"no result within {args.sharded_timeout} s: section abandoned, "
                                                         "the ranks left without tearing the process group down"}}
                    write_detail(fb)
                    print(compact_line(fb), flush=True)
                # a stalled collective is a failure of the run, not a success: the headline line is out, the status says
                # which part hung (every rank leaves with the same code; spawn_ranks hands it on)
                os._exit(EXIT_SHARDED_SECTION_HUNG)
            guard = threading.Timer(args.sharded_timeout, bail)
            guard.daemon = True
            guard.start()
        try:
            sharded = time_sharded_window(args, info, dist, backend, dev)
        except Exception as exc:   # noqa: BLE001 -- reported in the JSON line, the headline number stands
            sharded = {"error": f"{type(exc).__name__}: {exc}"}
        if guard is not None:
            guard.cancel()

    if info.rank == 0:
        counts = eng.counts()
        kf_per_s = D.whole_job_throughput(summaries, dt)
        # roofline of the Jacobian-evaluation kernel K1, measured live with HIP events on the
        # engine's own stream: algorithmic bytes of one launch / average launch duration
        alg_bytes = counts["imu"] * IMU_BYTES
        stages = {s: eng.time_stage(s, reps=5) for s in
                  ("linearize_imu", "linearize_between", "assemble", "assemble_idle", "solve", "retract", "decide")}
        solve_form = eng.solve_form()
        if solve_form == "assembling":
            # K3 is not part of this engine's LM trial (the forward sweep of `solve` forms H itself): what the stage timer
            # launched above is the stand-alone kernel, kept for comparison only
            stages["assemble_k3_not_in_the_step"] = stages.pop("assemble")
            stages.pop("assemble_idle")
            stages["assemble"] = 0.0
        prof = profiled_kernels()
        out = {
            "metric": "keyframes/sec fixed-lag update, 1k-pose window; ATE vs GTSAM ref",
            "value": kf_per_s, "unit": "keyframes/s", "n_gpus": info.world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"fixed-lag update of {args.window}-pose VIL windows (IMU + VIO + LiDAR between "
                                   f"factors at Carla rates), {args.iterations} LM trials per update, "
                                   f"{args.windows} independent windows per GPU",
                       "window_keyframes": args.window, "windows_per_gpu": args.windows,
                       "distinct_sequences_per_gpu": max(1, min(nseq, args.windows)),
                       "lm_trials_per_update": args.iterations,
                       "factors_per_gpu": {"imu": counts["imu"], "between": counts["between"]},
                       "parallelism": f"independent windows sharded over {info.world} rank(s), no data-path collective"},
            # (short objects first: a reader of the line's head sees the result, its accuracy and where the step's time goes)
            "accuracy": accuracy,
            "stage_ms": dict(stages, h2d=ingest_ms[0], preintegrate_tail=ingest_ms[1]),
            "roofline": roofline,
            "ingest": {"in_timed_step": True,
                       "what": "per update and window: the new keyframe's raw IMU samples (7 doubles each) + its 28-double between "
                               "record, one pinned host->device copy for all windows (stage_ms.h2d); K0 with each window's current "
                               "bias estimate (GraphManager.cpp:59) + staging of the between record in one launch "
                               "(stage_ms.preintegrate_tail); HIP events of the last timed update",
                       "bytes_per_update": int(feed[0][1].nbytes + feed[0][4].nbytes + feed[0][0].nbytes + feed[0][3].nbytes)},
            "lm_state_window0": lm_after_timed,
        }
        if prof is not None and "vf::k_linearize_imu" in prof.get("kernels", {}):
            # the committed rocprofv3 trace of this same command: the launch duration inside a whole step (other
            # kernels' traffic still draining), usually a few % above the isolated back-to-back figure
            pk = prof["kernels"]["vf::k_linearize_imu"]
            out["roofline"]["profiled"] = {"avg_launch_ms": pk["avg_ms"], "launches": pk["calls"], "source": prof["source"],
                                           "achieved": alg_bytes / (pk["avg_ms"] * 1e-3) / 1e9,
                                           "frac": alg_bytes / (pk["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS}
            # the same as flat keys (a parser that keeps only scalars of `roofline` still carries the committed trace's figure)
            out["roofline"].update({"profiled_avg_launch_ms": pk["avg_ms"], "profiled_launches": pk["calls"],
                                    "profiled_achieved": alg_bytes / (pk["avg_ms"] * 1e-3) / 1e9,
                                    "profiled_frac": alg_bytes / (pk["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                    "profiled_source": prof["source"]})
            if out["roofline"]["traffic"]:
                out["roofline"]["traffic_over_algorithmic"] = out["roofline"]["traffic"] / alg_bytes
        # the kernel that takes most of a step: K4 (banded Cholesky solve), HBM-bound under a full batch.  Algorithmic
        # bytes per keyframe: the band of H it needs (H[k][k-1]: 225, lower triangle of H[k][k]: 120, two 6x6 strips:
        # 432 doubles) + g + the panel written by the forward sweep and read back by the backward one (547 doubles
        # each way: 27 sub-diagonal rows + the rhs row x 15, and the 120 + 7 entries of L^-T its column pairs keep) + the
        # increment.
        k4_bytes_per_kf = 8 * (432 + 15 + 547 + 547 + 15)
        k4_name = "k_band_solve (K4: damped block-banded Cholesky factorisation + both substitutions)"
        if solve_form == "assembling":
            # the assembling sweep reads the Jacobians instead of H and g: the J stream (292 doubles per factor: the 291 entries
            # that are not structurally zero + padding), the residual (15), the between linearisation (78); panel, increment as above
            k4_bytes_per_kf = 8 * (292 + 15 + 78 + 547 + 547 + 15)
            from vil_sensor_fusion_amd import _lib as _vl
            _o = _vl.EngineTuningC()
            _vl.lib().vf_engine_default_tuning(_vl.C.byref(_o))
            asm_waves = args.solve_assemble_waves if args.solve_assemble_waves in (1, 2) else (2 if _o.solve_assemble_waves != 1 else 1)
            k4_name = (("k_band_forward_asm2 (two waves per window: eliminator + assembler on one LDS image)" if asm_waves == 2 else "k_band_forward_asm")
                       + " + k_band_backward (K3 + K4 in one pass: block rows of J^T J formed on the matrix cores "
                       "inside the forward sweep, damped block-banded Cholesky factorisation, both substitutions)")
        n_kf = args.windows * args.window
        k4_ach = n_kf * k4_bytes_per_kf / (stages["solve"] * 1e-3) / 1e9
        out["solve_form"] = solve_form
        out["roofline_solve"] = {"kernel": k4_name,
                                 "bound": "hbm", "achieved": k4_ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                 "frac": k4_ach / HBM_PEAK_GBPS, "avg_launch_ms": stages["solve"],
                                 "algorithmic_bytes_per_keyframe": k4_bytes_per_kf,
                                 "irreducible_io_bytes_per_keyframe": 8 * (432 + 15 + 15),
                                 "algorithmic_bytes_per_launch": n_kf * k4_bytes_per_kf,
                                 "frac_of_measured_copy_peak_6290": k4_ach / 6290.0,
                                 "note": "8.8 of the 12.4 KB per keyframe are the Cholesky panel written by the forward sweep and "
                                         "read back by the backward one; against H + g + delta alone (3.7 KB) the kernel moves 3.4x",
                                 "what_bounds_it": "not HBM: the per-window dependency chain at one wave per SIMD.  SQ counters "
                                                   "(profiles/*_sq_counters.md): the wave issues vector instructions 33 % and LDS "
                                                   "instructions 16 % of its cycles, is stalled on a dependent result 31 % and in "
                                                   "s_waitcnt 15 %; matrix pipes 11 % busy (DESIGN.md 7.9).  The HBM fraction is "
                                                   "quoted because it is what the kernel's traffic amounts to, not as its roofline",
                                 "traffic_note": "profiles/*_pmc_summary.md: k_band_solve read 2*FETCH_SIZE + WRITE_SIZE per launch"}
        if solve_form == "assembling":
            out["roofline_solve"]["irreducible_io_bytes_per_keyframe"] = 8 * (292 + 15 + 78 + 15)
            out["roofline_solve"]["note"] = ("8.8 of the 12.0 KB per keyframe are the Cholesky panel written by the forward sweep and read "
                                             "back by the backward one; the rest is the Jacobians K1 / K2 wrote (H is never stored)")
            out["roofline_solve"]["what_bounds_it"] = (
                "not HBM: the float64 units of a SIMD.  One window-step is the pivot chain of the 15 x 15 block (vector instructions "
                "at 7 cycles each) plus 26 v_mfma_f64_16x16x4 (12 of the Schur update, 14 that form the row of J^T J); a single wave "
                "gets one matrix instruction through every ~143 cycles, its vector instructions do not run under it, and the part's "
                "float64 matrix rate saturates at 47 TFLOP/s (tools/probes/mfma_f64_rate.hip); as two waves per window sharing the "
                "window's LDS the two jobs overlap each other's waits.  DESIGN.md 7.13, 7.15")
            out["roofline_solve"]["waves_per_window"] = asm_waves
            mfma_flop = 26 * 2048.0 * n_kf           # per launch of the forward sweep: 12 (Schur update) + 14 (rows of J^T J) tiles per keyframe
            out["roofline_solve"]["matrix_instruction_view"] = {
                "bound": "mfma", "unit": "TFLOP/s", "flop_per_launch": mfma_flop,
                "achieved": mfma_flop / (stages["solve"] * 1e-3) / 1e12,
                "peak_measured_one_wave_per_simd": 34.7, "peak_measured_two_waves_per_simd": 45.8, "peak_measured_saturated": 47.3,
                "peak_data_sheet_f64_matrix": 78.6,
                "frac_of_measured_peak_at_this_occupancy": mfma_flop / (stages["solve"] * 1e-3) / 1e12 / (45.8 if asm_waves == 2 else 34.7),
                "note": "the time is that of forward sweep + back substitution (the stage timer's `solve`); the back substitution "
                        "(0.92 ms of it) issues no matrix instruction; peaks measured by tools/probes/mfma_f64_rate.hip "
                        "(profiles/r04_mfma_f64_rate.log)"}
        # The issue view: the forward sweep is bound by what one SIMD can ISSUE in float64, not by HBM.  From
        # the committed counter passes of this same command (profiles/*_sq_counters.md -> kernel_durations.json): every wave64
        # vector instruction occupies the SIMD's 16 lanes for 4 cycles (float64 FMA at full rate: 78.6 TFLOP/s = 16 lanes x 2 x
        # 1024 SIMDs x 2.4 GHz), a v_mfma_f64_16x16x4 (2048 flop) the matrix pipe for 64 cycles at the data-sheet rate and 106
        # at the rate this part saturates at (47.3 TFLOP/s, tools/probes/mfma_f64_rate.hip); a SIMD has launch x 2.4e9 cycles.
        fwd_key = next((k for k in ("vf::k_band_forward_asm2", "vf::k_band_forward_asm", "vf::k_band_solve") if prof and k in prof.get("kernels", {})), None)
        if fwd_key and "sq" in prof["kernels"][fwd_key] and "valu_insts_per_launch" in prof["kernels"][fwd_key]["sq"]:
            pk, sqc = prof["kernels"][fwd_key], prof["kernels"][fwd_key]["sq"]
            simd_cycles = pk["full_avg_ms"] * 1e-3 * 2.4e9 * 1024 / n_kf
            valu_c = (sqc["valu_insts_per_launch"] - sqc["mfma_f64_insts_per_launch"]) * 4.0 / n_kf
            f64_c = sqc["valu_f64_insts_per_launch"] * 4.0 / n_kf
            mf = sqc["mfma_f64_insts_per_launch"] / n_kf
            out["roofline_solve"]["issue_view"] = {
                "kernel": fwd_key, "bound": "float64 issue of a SIMD (vector unit + matrix pipe)", "unit": "SIMD cycles per keyframe",
                "available": simd_cycles, "vector_instructions_issue": valu_c, "of_which_float64": f64_c,
                "matrix_instructions_per_keyframe": mf, "matrix_pipe_at_data_sheet_rate": mf * 64.0, "matrix_pipe_at_measured_saturation": mf * 2048.0 / (47.3e12 / 1024 / 2.4e9),
                "matrix_pipe_busy_measured": sqc["mfma_busy_cycles_per_launch"] / n_kf,
                "frac_issue_serial": (valu_c + mf * 2048.0 / (47.3e12 / 1024 / 2.4e9)) / simd_cycles,
                "frac_issue_overlapped": max(valu_c, mf * 2048.0 / (47.3e12 / 1024 / 2.4e9)) / simd_cycles,
                "note": "frac_issue_serial: vector and matrix work of a SIMD one after the other (what ONE wave per SIMD can do: its "
                        "vector instructions do not issue under its own matrix instruction); frac_issue_overlapped: both units busy "
                        "at once (two waves per SIMD, perfectly interleaved).  The kernel sits between the two; the rest is the pivot "
                        "chain's dependent-result latency (wait / issue-stall columns of the counter table)",
                "source": prof["source"]}
        if prof is not None:
            out["profiled_kernels"] = prof
        if conv is not None:
            out["with_convergence_exit"] = conv
        if incr is not None:
            out["incremental_update"] = incr
        if sharded is not None:
            out["time_sharded_window"] = sharded
        if not args.no_single_window:
            # latency of the same update on ONE window (what a single vehicle sees)
            sv = argparse.Namespace(**vars(args))
            one, one_feed = make_engine(sv, gpu, 1, one_seq, one_updates)
            fed[id(one)] = 0            # (ids of closed engines can be reused: every engine starts its feed explicitly)
            for _ in range(2):
                one_step(one, one_feed)
            one.sync()
            t1 = time.perf_counter()
            for _ in range(5):
                one_step(one, one_feed)
            one.sync()
            lat = (time.perf_counter() - t1) / 5
            out["single_window"] = {"ms_per_update": lat * 1e3, "keyframes_per_s": 1.0 / lat,
                                    "solve_ms": one.time_stage("solve", reps=5)}
            one.close()
            # the same with vf_engine_opts.use_hip_graph (the launch sequence of iterate replayed from a captured hipGraph;
            # bit-identical, tests/test_gpu_hip_graph.py): what replay does to the one-window latency
            try:
                g1, g1_feed = make_engine(sv, gpu, 1, one_seq, one_updates, use_hip_graph=True)
                fed[id(g1)] = 0
                for _ in range(2):
                    one_step(g1, g1_feed)
                g1.sync()
                t1 = time.perf_counter()
                for _ in range(5):
                    one_step(g1, g1_feed)
                g1.sync()
                en, cap, rep = g1.graph_info()
                out["single_window"]["with_hip_graph"] = {"ms_per_update": (time.perf_counter() - t1) / 5 * 1e3, "replay_active": en,
                                                          "captures": cap, "replays": rep}
                g1.close()
            except Exception as exc:   # noqa: BLE001
                out["single_window"]["with_hip_graph"] = {"error": f"{type(exc).__name__}: {exc}"}
        if not args.no_degeneracy and info.world == 1:
            out["degeneracy_k6"] = degeneracy_section()
        if not args.no_graph_manager and info.world == 1:
            out["graph_manager"] = graph_manager_section()
        if cpu_one is not None:
            out["cpu_baseline"] = cpu_one
            if cpu_all is not None:
                out["cpu_baseline_all_cores"] = cpu_all
                out["cpu_baseline_all_cores"]["gpu_over_cpu_node"] = kf_per_s / cpu_all["value"]
        write_detail(out)
        line = compact_line(out)
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
