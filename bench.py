#!/usr/bin/env python3
"""Headline benchmark: keyframes/sec of fixed-lag smoother updates on 1000-pose windows.

One *step* = one fixed-lag update of every window in the batch: append one keyframe (its IMU
factor + between factor are already resident in HBM), drop the oldest one (re-anchor the prior),
then K Levenberg-Marquardt trials, each = linearise ALL factors of the window (K1+K2) ->
block-banded J^T J (K3) -> banded Cholesky solve (K4) -> retract + cost + accept/reject (K5).
value = windows * steps / time  (one new keyframe per window per step), whole job over all ranks.

Contract: python bench.py --gpus N --steps K --warmup W ; rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

IMU_BYTES = 5496      # 1776 B read + 3720 B written per IMU factor (unfused)
BTW_BYTES = 960       # 336 + 624 per between factor
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def cpu_baseline(args):
    """The CPU oracle (a port, not GTSAM: GTSAM cannot be built here) doing the same update on a
    bounded sample: one window, `cpu_steps` updates of K LM trials each."""
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
        win.lm(iterations=args.iterations, n_threads=1)
        prob["states"][s:s + n] = win.states
    dt = time.perf_counter() - t0
    return dict(value=args.cpu_steps / dt, unit="keyframes/s", cores=1, kind="port",
                sample=f"{args.cpu_steps} fixed-lag updates of one {n}-pose window, {args.iterations} LM trials each, "
                       f"single-thread C restatement (oracle/vf_oracle.c); reference CPU GTSAM cannot be built or timed here")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--window", type=int, default=1000, help="keyframes per window (BASELINE metric: 1000)")
    ap.add_argument("--windows", type=int, default=1024, help="independent windows per GPU")
    ap.add_argument("--sequences", type=int, default=8, help="distinct synthetic sequences per rank")
    ap.add_argument("--iterations", type=int, default=5, help="LM trials per update")
    ap.add_argument("--cpu-steps", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from vil_sensor_fusion_amd import Engine, EngineOpts, synth
    from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS

    def make(rank):
        n, total = args.window, args.window + args.steps + args.warmup + 1
        eng = Engine(EngineOpts(windows=args.windows, capacity=total, device=local_rank))
        nseq = min(args.sequences, args.windows)
        seqs = [synth.make_sequence(seed=1000 * rank + s, n_kf=total) for s in range(nseq)]
        recs = [synth.between_records(s) for s in seqs]
        for w in range(args.windows):
            seq = seqs[w % nseq]
            gt0 = seq.gt_states[0]
            eng.preintegrate(w, 1, seq.imu_off[1:], seq.imu_steps, np.zeros(6), synth.CARLA_IMU_COV)
            eng.set_between(w, seq.btw_a, seq.btw_b, recs[w % nseq])
            eng.set_states(w, 0, gt0.reshape(1, 16))
            eng.set_prior(w, 0, synth.prior_record(gt0, REFERENCE_PRIOR_SIGMAS))
            eng.set_range(w, 0, 1)
            eng.predict(w, 1, n - 1)
            eng.set_range(w, 0, n)
        eng.sync()
        return eng

    eng = make(rank)
    eng.iterate(args.iterations)      # converge the initial windows (not timed)
    eng.sync()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        eng.sync()

    def one_step():
        eng.slide(REFERENCE_PRIOR_SIGMAS)
        eng.iterate(args.iterations)

    for _ in range(args.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    out = None
    if rank == 0:
        counts = eng.counts()
        # roofline of the Jacobian-evaluation kernel (K1), measured live with HIP events on the
        # engine's stream: algorithmic bytes of one launch / average launch duration
        k1_ms = eng.time_stage("linearize_imu", reps=20)
        k2_ms = eng.time_stage("linearize_between", reps=20)
        alg_bytes = counts["imu"] * IMU_BYTES
        achieved = alg_bytes / (k1_ms * 1e-3) / 1e9
        stages = {s: eng.time_stage(s, reps=5) for s in ("assemble", "solve", "retract", "decide")}
        stages["linearize_imu"], stages["linearize_between"] = k1_ms, k2_ms
        lm = eng.read_lm(0)
        kf_per_s = args.windows * world * args.steps / dt
        out = {
            "metric": "keyframes/sec fixed-lag update, 1k-pose window; ATE vs GTSAM ref",
            "value": kf_per_s, "unit": "keyframes/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"fixed-lag update of {args.window}-pose VIL windows (IMU + VIO + LiDAR between "
                                   f"factors, Carla rates), {args.iterations} LM trials per update, "
                                   f"{args.windows} independent windows per GPU",
                       "window_keyframes": args.window, "windows_per_gpu": args.windows,
                       "lm_trials_per_update": args.iterations,
                       "factors_per_gpu": {"imu": counts["imu"], "between": counts["between"]},
                       "parallelism": f"replicated windows x{world} (no data-path collective)"},
            "roofline": {"kernel": "k_linearize_imu (K1, Jacobian evaluation)", "bound": "hbm",
                         "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": k1_ms},
            "stage_ms": stages,
            "lm_state_window0": lm,
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
            out["cpu_baseline"]["gpu_over_cpu"] = kf_per_s / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
