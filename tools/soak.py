"""Soak: 10^4 fixed-lag updates of (a) a batch engine in the headline's solver form (the two-wave
assembling sweep, forced at this batch size) with compaction cycles, and (b) a GraphManager with a 1 000-keyframe lag fed
like the node -- one keyframe, one between factor, one vf_solve at a time -- both against the CPU oracle doing the same
updates, compared every `--check-every` updates.  A 20 Hz node does 10^4 updates in eight minutes.

The oracle chains for the engine's compared windows run in forked workers started BEFORE this process touches the GPU (each
writes its checkpoints to a file); the GraphManager's oracle mirror runs in this process, interleaved with the device's
solves, on the IMU records the device preintegrated (K0 parity is the business of tests/test_gpu_k0_covariance.py).

    python tools/soak.py --updates 10000 --windows 256 --check-every 500 > profiles/r05_soak_10k_updates.txt
"""
import argparse
import multiprocessing
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def oracle_chain(job):
    """FixedLagOracle of one sequence: states of the window at every checkpoint -> npz"""
    seed, n, updates, every, K, init_iters, path, refine, perturb = job
    from oracle import oracle
    from tests import helpers
    from vil_sensor_fusion_amd import synth
    oracle.build()
    seq = synth.make_sequence(seed=seed, n_kf=n + updates + 2)
    prob = helpers.build_problem(oracle, seq)
    if perturb > 0.0:
        # the same chain on inputs that differ in the last place: every between measurement (rotation and translation) times
        # (1 + perturb * N(0, 1)) -- what two float64 implementations of the same update differ by, made explicit
        rng = np.random.default_rng(12345)
        prob["btw"] = prob["btw"].copy()
        prob["btw"][:, :7] *= 1.0 + perturb * rng.normal(size=prob["btw"][:, :7].shape)
    ref = helpers.FixedLagOracle(oracle, prob, n, K, init_iterations=init_iters, ingest=(seq, oracle.carla_imu_params()), refine=refine)
    out = {"u0": ref.window_states.copy()}
    t0 = time.time()
    for u in range(1, updates + 1):
        ref.update()
        if u % every == 0:
            out[f"u{u}"] = ref.window_states.copy()
            np.savez(path, **out)
            print(f"[oracle seed {seed}] update {u}: cost {ref.costs[-1]:.6f}, {time.time() - t0:.0f} s", flush=True)
    np.savez(path, **out)
    return path


class GraphOracle:
    """the CPU oracle doing what vf_solve does with a lag (vf_graph.cpp: marginalise the keyframes that fall out of the lag at
    the linearisation of the previous solve, K LM trials on the rest), one keyframe per solve, on the records the device holds"""

    def __init__(self, oracle, seq, lag, K, total):
        from vil_sensor_fusion_amd import synth
        from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS
        self.o, self.seq, self.lag, self.K = oracle, seq, lag, K
        self.states = np.zeros((total, 16))
        self.states[0] = seq.gt_states[0]
        self.imu = np.zeros((total, 190))
        self.btw = synth.between_records(seq)
        self.prior = synth.prior_record(seq.gt_states[0], REFERENCE_PRIOR_SIGMAS)
        self.g = np.array([0.0, 0.0, -9.81])
        self.s, self.hi, self.marg, self.win, self.solved = 0, 1, None, None, 0

    def step(self, rec190):
        k = self.hi
        self.imu[k] = rec190
        self.states[k] = self.o.predict(rec190, self.g, self.states[k - 1])
        self.hi += 1
        if self.hi - self.s > self.lag and self.solved - self.s >= 3:
            self.marg = self.win.marginalize(0, self.o.prior_gauge_floor(self.win.n_kf))          # (one keyframe per solve: at most one leaves)
            self.marg.k0 = 0
            self.s += 1
            assert self.hi - self.s <= self.lag
        lo, hi, seq = self.s, self.hi, self.seq
        m = (seq.btw_a >= lo) & (seq.btw_b < hi)
        ks = np.arange(lo + 1, hi)
        with_prior = self.marg is None
        pk = np.array([0], dtype=np.int32) if with_prior else np.zeros(0, dtype=np.int32)
        pd = self.prior.reshape(1, -1) if with_prior else np.zeros((0, 31))
        w = self.o.Window(self.states[lo:hi], ks - 1 - lo, ks - lo, self.imu[lo + 1:hi], seq.btw_a[m] - lo, seq.btw_b[m] - lo,
                          self.btw[m], pk, pd, self.g)
        if self.marg is not None:
            w.set_marg(self.marg)
        self.costs, self.acc, _ = w.lm(iterations=self.K)
        self.states[lo:hi] = w.states
        self.win, self.solved = w, hi - 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--updates", type=int, default=10000)
    ap.add_argument("--graph-updates", type=int, default=-1, help="vf_solve calls of the GraphManager part (default: --updates)")
    ap.add_argument("--windows", type=int, default=256)
    ap.add_argument("--window", type=int, default=1000)
    ap.add_argument("--sequences", type=int, default=16, help="distinct synthetic sequences (window w runs sequence w mod this)")
    ap.add_argument("--compare", type=int, default=4, help="windows (= sequences) followed by an oracle chain")
    ap.add_argument("--check-every", type=int, default=500)
    ap.add_argument("--iterations", type=int, default=5)
    ap.add_argument("--init-iterations", type=int, default=200)
    ap.add_argument("--head-room", type=int, default=128, help="engine keyframe slots beyond the window (a compaction whenever they run out)")
    ap.add_argument("--form", default="assembling2", choices=["assembling2", "assembling1", "two_kernel", "default"], help="solver form of the batch engine")
    ap.add_argument("--refine", type=int, default=0, help="corrections through J after every solve, device and oracle alike (vf_engine_opts.refine_iterations)")
    ap.add_argument("--skip-engine", action="store_true")
    ap.add_argument("--skip-graph", action="store_true")
    ap.add_argument("--out-dir", default=os.path.join(ROOT, "gpurun_out"))
    args = ap.parse_args()
    gupdates = args.updates if args.graph_updates < 0 else args.graph_updates
    n, U, K = args.window, args.updates, args.iterations
    os.makedirs(args.out_dir, exist_ok=True)
    from oracle import oracle
    oracle.build()
    from tests import helpers
    from vil_sensor_fusion_amd import synth

    # ---- oracle chains of the compared windows: forked before the GPU is touched
    ctx = multiprocessing.get_context("fork")
    pool, pending = None, []
    if not args.skip_engine:
        pool = ctx.Pool(args.compare + 1)
        jobs = [(s, n, U, args.check_every, K, args.init_iterations, os.path.join(args.out_dir, f"soak_oracle_{s}.npz"), args.refine, 0.0) for s in range(args.compare)]
        # ... and one more chain of window 0 on inputs perturbed in the last place (1e-15 relative): the oracle against itself
        jobs.append((0, n, U, args.check_every, K, args.init_iterations, os.path.join(args.out_dir, "soak_oracle_0_perturbed.npz"), args.refine, 1e-15))
        pending = [pool.apply_async(oracle_chain, (j,)) for j in jobs]
    t_all = time.time()
    seqs = [synth.make_sequence(seed=s, n_kf=n + max(U, gupdates) + 2, keep_raw=(s == 0)) for s in range(args.sequences)]
    print(f"# soak: {U} updates of {args.windows} windows x {n} keyframes ({args.sequences} sequences), K = {K}; GraphManager: {gupdates} solves at lag {n}; "
          f"sequences made in {time.time() - t_all:.0f} s", flush=True)

    from vil_sensor_fusion_amd import Engine, EngineOpts
    from vil_sensor_fusion_amd.engine import REFERENCE_PRIOR_SIGMAS
    from vil_sensor_fusion_amd.graph_manager import GraphManager

    # ================================================================ (b) GraphManager, fed like the node
    if not args.skip_graph:
        seq = seqs[0]
        cap = n + 192
        gm = GraphManager(capacity=cap, lag=n, iterations=K, rel_tol=0.0, abs_tol=0.0)
        gm.setInitialState(seq.gt_states[0])
        ref = GraphOracle(oracle, seq, n, K, gupdates + 2)
        cov = {c: np.eye(6) * c for c in (synth.VIO_COV, synth.LIDAR_COV)}
        by_end = {int(b): i for i, b in enumerate(seq.btw_b)}
        gm.addIMUMeasurement(0.0, seq.imu_acc[0], seq.imu_gyro[0])
        i_imu, worst, worst_al, t0, t_gpu, last_print = 0, 0.0, 0.0, time.time(), 0.0, time.time()
        print("## GraphManager (vf_solve per keyframe, lag 1000, compaction whenever the slots run out)", flush=True)
        for k in range(1, gupdates + 1):
            t_k = seq.kf_time[k]
            while i_imu < seq.imu_t.size and seq.imu_t[i_imu] <= t_k + 0.011:
                gm.addIMUMeasurement(seq.imu_t[i_imu], seq.imu_acc[i_imu], seq.imu_gyro[i_imu])
                i_imu += 1
            key = gm.reserveNode(t_k)
            assert key == k
            if k in by_end:
                i = by_end[k]
                gm.addBetweenFactor(int(seq.btw_a[i]), k, (seq.btw_q[i], seq.btw_t[i]), cov[float(seq.btw_cov[i])])
            tg = time.perf_counter()
            gm.solve()
            t_gpu += time.perf_counter() - tg
            ref.step(gm.imuFactor(k))
            if k % args.check_every == 0 or k == gupdates:
                lo = max(ref.s, k - n + 1)
                traj = gm.trajectory(lo, k - lo + 1)
                a, r = helpers.ate(traj, ref.states[lo:k + 1])
                al, psi, _ = helpers.ate_gauge_aligned(traj, ref.states[lo:k + 1])
                worst, worst_al = max(worst, a), max(worst_al, al)
                lm = gm.lmStats()
                print(f"solve {k:6d}: window keys [{ref.s}, {k}] vs oracle ATE {a:.3e} m (gauge-aligned {al:.1e} m, yaw {psi:+.1e} rad) rot {r:.3e} rad; cost {lm['cost']:.9f} (oracle {ref.costs[-1]:.9f}); "
                      f"accepted {lm['accepted']} rejected {lm['rejected']} failed solves {lm['solve_failures']}; vf_solve mean {t_gpu / k * 1e3:.2f} ms; {time.time() - t0:.0f} s", flush=True)
                assert lm["solve_failures"] == 0
                last_print = time.time()
            elif time.time() - last_print > 60:
                print(f"   ... solve {k}", flush=True)
                last_print = time.time()
        print(f"GraphManager soak: {gupdates} solves, no failed solve, worst ATE vs the oracle at a checkpoint {worst:.3e} m unaligned, {worst_al:.3e} m gauge-aligned", flush=True)
        gm.close()

    # ================================================================ (a) batch engine, headline solver form
    if not args.skip_engine:
        B, nseq = args.windows, args.sequences
        cap = n + args.head_room             # the window + two tiles of head room -> a compaction every 64..128 updates
        form = {"assembling2": dict(sweep_two_sided_max=0, chunks=1, solve_assemble_min=1, solve_assemble_waves=2),
                "assembling1": dict(sweep_two_sided_max=0, chunks=1, solve_assemble_min=1, solve_assemble_waves=1),
                "two_kernel": dict(sweep_two_sided_max=0, chunks=1, solve_assemble_min=0), "default": {}}[args.form]
        eng = Engine(EngineOpts(windows=B, capacity=cap, refine_iterations=args.refine, lm_excursion=0, **form))
        recs = [synth.between_records(s) for s in seqs]
        for w in range(B):
            seq = seqs[w % nseq]
            eng.preintegrate(w, 1, seq.imu_off[1:n + 1], seq.imu_steps, np.zeros(6), synth.CARLA_IMU_COV)
            m = seq.btw_b < n
            eng.set_between(w, seq.btw_a[m], seq.btw_b[m], recs[w % nseq][m])
            eng.set_states(w, 0, seq.gt_states[0].reshape(1, 16))
            eng.set_prior(w, 0, synth.prior_record(seq.gt_states[0], REFERENCE_PRIOR_SIGMAS))
            eng.set_range(w, 0, 1)
        eng.predict(-1, 1, n - 1)
        for w in range(B):
            eng.set_range(w, 0, n)
        eng.iterate(args.init_iterations)
        eng.sync()
        print(f"## engine: {B} windows, form {eng.solve_form()}, capacity {cap} slots", flush=True)
        by_end = [{int(b): i for i, b in enumerate(s.btw_b)} for s in seqs]
        none_rec = np.zeros(28)
        base, hi, compactions, worst, worst_al, t0, last_print = 0, n, 0, 0.0, 0.0, time.time(), time.time()
        oracle_files = {}
        for u in range(1, U + 1):
            k = n + u - 1                                   # keyframe appended by this update
            if hi - base >= cap:                            # no free slot: reclaim whole tiles below the window
                shift = ((k - n) - base) // 64 * 64         # lo = k - n (in keyframes) before this update's slide
                eng.compact(shift)
                base += shift
                compactions += 1
            off, steps, a, rec = [0], [], [], []
            for w in range(B):
                s = w % nseq
                seq = seqs[s]
                st = seq.imu_steps[seq.imu_off[k]:seq.imu_off[k + 1]]
                steps.append(st)
                off.append(off[-1] + st.shape[0])
                i = by_end[s].get(k, -1)
                a.append(int(seq.btw_a[i]) - base if i >= 0 else -1)
                rec.append(recs[s][i] if i >= 0 else none_rec)
            eng.ingest_tail(np.array(off, dtype=np.int32), np.concatenate(steps), synth.CARLA_IMU_COV, np.array(a, dtype=np.int32), np.array(rec))
            eng.slide(REFERENCE_PRIOR_SIGMAS, marginalize=True)
            eng.iterate(K)
            hi += 1
            if u % args.check_every == 0 or u == U:
                eng.ingest_status()
                fails = sum(eng.read_lm(w)["solve_failures"] for w in range(B))
                line = []
                for c in range(args.compare):
                    path = os.path.join(args.out_dir, f"soak_oracle_{c}.npz")
                    for _ in range(3600):                  # the oracle chain may be behind the device: wait for its checkpoint
                        try:
                            F = np.load(path)
                            if f"u{u}" in F.files:
                                break
                        except (OSError, ValueError, EOFError):
                            pass
                        time.sleep(1.0)
                        if time.time() - last_print > 60:
                            print(f"   ... waiting for the oracle's checkpoint {u} of window {c}", flush=True)
                            last_print = time.time()
                    x = eng.get_states(c, k - n + 1 - base, n)
                    at, rt = helpers.ate(x, F[f"u{u}"])
                    al, psi, _ = helpers.ate_gauge_aligned(x, F[f"u{u}"])
                    worst, worst_al = max(worst, at), max(worst_al, al)
                    line.append(f"{at:.2e} (gauge-aligned {al:.1e}, yaw {psi:+.1e} rad)")
                lm0 = eng.read_lm(0)
                try:
                    P = np.load(os.path.join(args.out_dir, "soak_oracle_0_perturbed.npz"))
                    self_d = f"{helpers.ate(P[f'u{u}'], np.load(os.path.join(args.out_dir, 'soak_oracle_0.npz'))[f'u{u}'])[0]:.2e}" if f"u{u}" in P.files else "pending"
                except (OSError, ValueError, EOFError, KeyError):
                    self_d = "pending"
                line.append(f"oracle vs itself on 1e-15-perturbed inputs (window 0): {self_d}")
                print(f"update {u:6d}: ATE vs oracle per compared window [{', '.join(line)}] m; window 0 cost {lm0['cost']:.9f} accepted {lm0['accepted']} "
                      f"rejected {lm0['rejected']}; failed solves over all windows {fails}; compactions {compactions}; {time.time() - t0:.0f} s", flush=True)
                assert fails == 0
                last_print = time.time()
            elif time.time() - last_print > 60:
                print(f"   ... update {u}", flush=True)
                last_print = time.time()
        print(f"engine soak: {U} updates x {B} windows = {U * B} window-updates, {compactions} compactions, no failed solve, "
              f"worst ATE vs the oracle at a checkpoint {worst:.3e} m unaligned, {worst_al:.3e} m with the window's global translation / yaw "
              f"(a gauge no factor sees) fitted out", flush=True)
        eng.close()
        for p in pending:
            p.get(timeout=3600)
        pool.close()
    print(f"# soak done in {time.time() - t_all:.0f} s")


if __name__ == "__main__":
    main()
