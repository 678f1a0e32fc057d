"""One process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" in CPU tests).

The hot path shards over independent smoothing windows (vehicles / sequences): every rank owns a
contiguous block of windows and runs the whole K0-K5 pipeline on its own MI355X; there is NO
data-path collective.  The only communication is control-plane: a barrier around the timed region,
a MAX-reduce of the elapsed time and an all-gather of per-rank summaries.

ShardedSolver is the other multi-GPU form (BASELINE.json configs[4]): ONE window spread
in time over the ranks.  Every rank holds the window, owns a contiguous range of the chunks of the
partitioned solve (K4p) and the keyframes they cover, and an LM trial has TWO collectives: an in-place
all-gather of the packed separator system (one buffer, 2248 doubles per chunk and window) and an
all-reduce of the increments (15 doubles per keyframe slot) with the solve-failure flags behind them.
The cost of a trial needs none: every rank evaluates every residual (Jacobians only for its own
factors) and takes the same accept / reject decision.  RCCL collectives are enqueued on the stream the
engine's kernels run on, so a trial needs no host synchronisation.
"""
from __future__ import annotations

import os
from dataclasses import dataclass


@dataclass
class RankInfo:
    rank: int
    local_rank: int
    world: int


def rank_info() -> RankInfo:
    return RankInfo(int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
                    int(os.environ.get("WORLD_SIZE", "1")))


def shard_windows(total_windows: int, rank: int, world: int):
    """Contiguous, balanced split of window ids [0,total) -> [lo,hi) for `rank`."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, extra = divmod(total_windows, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def init(backend: str | None = None, device_id=None):
    """Initialise the default process group when WORLD_SIZE > 1; returns torch.distributed or None."""
    info = rank_info()
    if info.world <= 1 and not os.environ.get("VF_FORCE_DIST"):   # VF_FORCE_DIST: 1-rank smoke test of the backend
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not dist.is_initialized():
        kw = {}
        if device_id is not None:
            kw["device_id"] = device_id
        dist.init_process_group(backend or "nccl", **kw)
    return dist


def barrier(dist):
    if dist is not None:
        dist.barrier()


def max_over_ranks(dist, value: float, device="cpu") -> float:
    if dist is None:
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_summaries(dist, summary: dict):
    """all_gather_object of small per-rank dicts (window range, keyframes processed, final costs)."""
    if dist is None:
        return [summary]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, summary)
    return out


def whole_job_throughput(summaries, seconds: float) -> float:
    """value = units all ranks processed / max-over-ranks time."""
    return sum(s["keyframes"] for s in summaries) / seconds


# ------------------------------------------------------------------------------------------------
# time-sharded windows
def chunk_geometry(n: int, chunks: int, fit: bool = False):
    """[(first, interior, has_separator)] of the partitioned solve (libvilfusion's own geometry)."""
    import ctypes as C
    from . import _lib
    l = _lib.lib()
    cnt = C.c_int()
    _lib.check(l.vf_chunk_geometry(n, chunks, int(fit), -1, C.byref(cnt), None, None, None))
    out = []
    for c in range(cnt.value):
        a, b, h = C.c_int(), C.c_int(), C.c_int()
        _lib.check(l.vf_chunk_geometry(n, chunks, int(fit), c, None, C.byref(a), C.byref(b), C.byref(h)))
        out.append((a.value, b.value, bool(h.value)))
    return out


def shard_range(n: int, chunks: int, rank: int, world: int, fit: bool = False):
    """(chunk_lo, chunk_hi, kf_lo, kf_hi) owned by `rank`."""
    import ctypes as C
    from . import _lib
    v = [C.c_int() for _ in range(4)]
    _lib.check(_lib.lib().vf_shard_range(n, chunks, int(fit), rank, world, *[C.byref(x) for x in v]))
    return tuple(x.value for x in v)


def probe_inplace_all_gather(dist, device, rank: int, world: int) -> bool:
    """Does this torch / RCCL build accept an all-gather whose send buffer is the rank's own slice of the receive buffer?
    Decided ONCE, at set-up, on every rank alike, with a four-double probe: a build that refuses aliased buffers raises
    while checking its arguments -- before anything is communicated, identically on all ranks -- so every rank reaches the
    same answer and the hot path never has to interpret an error message."""
    import torch
    probe = torch.zeros(4 * world, dtype=torch.float64, device=device)
    probe[4 * rank:4 * rank + 4] = float(rank + 1)
    try:
        dist.all_gather_into_tensor(probe, probe[4 * rank:4 * rank + 4])
    except (RuntimeError, ValueError):
        return False
    torch.cuda.synchronize(device)
    want = torch.arange(1, world + 1, dtype=torch.float64, device=device).repeat_interleave(4)
    return bool(torch.equal(probe, want))


def all_gather_slices(dist, full, per_slice: int, rank: int, world: int, backend: str, inplace: bool = True):
    """`full` = world equal slices of per_slice elements; rank r holds slice r; afterwards all hold all.
    nccl (RCCL): ONE all-gather on the device, in place when `inplace` (the send buffer is the rank's own slice of the
    receive buffer: no clone, no temporary, no copy back; `probe_inplace_all_gather` says whether the build takes that),
    otherwise from a copy of the rank's slice.  gloo (CPU tests, shared-GPU smoke runs): staged through the host.
    An error of the collective itself is never retried: a second collective issued by one rank alone would hang the others."""
    if dist is None or (world == 1 and backend != "nccl"):
        return
    own = full[rank * per_slice:(rank + 1) * per_slice]
    if backend == "nccl":
        dist.all_gather_into_tensor(full[:world * per_slice], own if inplace else own.clone())
        return
    import torch
    parts = [torch.empty(per_slice, dtype=full.dtype) for _ in range(world)]
    dist.all_gather(parts, own.detach().cpu().contiguous())
    full[:world * per_slice].copy_(torch.cat(parts))


def all_reduce_sum(dist, t, backend: str):
    if dist is None:
        return
    if backend == "nccl":
        dist.all_reduce(t)
        return
    c = t.detach().cpu()
    dist.all_reduce(c)
    t.copy_(c)


class _DevicePtr:
    """Raw device pointer -> torch tensor without a copy (__cuda_array_interface__)."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


class ShardedSolver:
    """LM on windows spread in time over the ranks of `dist` (see the module docstring).

    eng: an Engine created on every rank with the same explicit `chunks` (a multiple of the world size)
    and loaded with the same window(s).  iterate(K) has the semantics of Engine.iterate(K)."""

    def __init__(self, eng, dist, device, backend: str = "nccl"):
        import torch
        self.eng, self.dist, self.backend = eng, dist, backend
        self.rank = dist.get_rank() if dist is not None else 0
        self.world = dist.get_world_size() if dist is not None else 1
        self.device = torch.device(device)
        eng.set_shard(self.rank, self.world)
        # kernels and collectives share torch's current stream: no host round trip inside a trial
        eng.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        info = eng.shard_info()
        wrap = lambda p, n: torch.as_tensor(_DevicePtr(p, n), device=self.device)
        self.chunks = info.chunks
        self.per_rank = info.chunks // self.world
        self.sep_per = info.sep_per_chunk
        self.sep = wrap(info.sep, info.chunks * self.sep_per)
        self.delta = wrap(info.delta, info.delta_count)       # increments + one solve-failure flag per window
        # refined solve (vf_engine_opts.refine_iterations): the correction solves leave their increments in a buffer of their own
        self.refine_delta = wrap(info.refine_delta, info.delta_count) if info.refine_delta else None
        self.collectives = 0                                   # issued so far (0 when there is no process group)
        # whether the in-place form of the separator all-gather is usable is settled here, once, on every rank alike
        self.inplace_all_gather = (probe_inplace_all_gather(dist, self.device, self.rank, self.world)
                                   if (dist is not None and backend == "nccl") else True)

    COLLECTIVES_PER_TRIAL = 2

    def _count(self):
        if self.dist is not None:
            self.collectives += 1

    # one LM trial = three device phases with a collective between them (LockstepGroup drives the same phases for
    # several shards held by ONE process)
    def phase_local(self):
        self.eng.assemble()
        self.eng.solve_local()

    def exchange_sep(self):
        all_gather_slices(self.dist, self.sep, self.sep_per * self.per_rank, self.rank, self.world, self.backend,
                          inplace=self.inplace_all_gather)
        self._count()

    def phase_global(self):
        self.eng.solve_global()

    def exchange_delta(self):
        all_reduce_sum(self.dist, self.delta, self.backend)
        self._count()

    def exchange_refine(self):
        all_reduce_sum(self.dist, self.refine_delta, self.backend)
        self._count()

    def refine(self):
        """the refined solve, staged: every correction is one more solve with the trial's factor shape (two collectives) and two
        passes over the Jacobians, which every rank holds in full; the count is the same on every rank"""
        e = self.eng
        n = e.refine_count()
        if not n:
            return
        e.refine_begin()
        for _ in range(n):
            e.solve_local()
            self.exchange_sep()
            e.solve_global()
            self.exchange_refine()
            e.refine_step()
        e.refine_end()

    def phase_finish(self):
        e = self.eng
        e.retract()
        e.linearize(1)
        e.decide(False)

    def begin(self):
        e = self.eng
        e.reset_lambda()
        e.linearize(0)
        e.decide(True)

    def trial(self):
        self.phase_local()
        self.exchange_sep()
        self.phase_global()
        self.exchange_delta()
        self.refine()
        self.phase_finish()

    def gn_step(self, relin_threshold=1e-4):
        """one reference-compat update (vf_engine_isam_step) of the time-sharded window: undamped Gauss-Newton about the
        linearisation points, the estimate lands in the trial buffer (Engine.get_estimate)"""
        self.eng.gn_begin(relin_threshold)
        self.phase_local()
        self.exchange_sep()
        self.phase_global()
        self.exchange_delta()
        self.refine()
        self.eng.retract()

    def iterate(self, iterations: int):
        self.begin()
        for _ in range(iterations):
            self.trial()
        self.eng.close_excursions()


class _LocalRank:
    """The two methods of torch.distributed ShardedSolver's constructor asks for, for a shard that lives in this process."""

    def __init__(self, rank, world):
        self._r, self._w = rank, world

    def get_rank(self):
        return self._r

    def get_world_size(self):
        return self._w


class LockstepGroup:
    """`world` shards of ONE time-sharded window held by one process on one GPU: the geometry of an N-GPU run (BASELINE
    configs[4]: 8 ranks x 12 chunks of a 10 000-pose window) where only one GPU -- and at most six GPU processes -- is to
    be had.  Every shard is a real engine with vf_engine_set_shard(r, world) running ShardedSolver's own phases; the two
    collectives of a trial are carried out on the device between the engines' buffers (slice copies for the all-gather,
    a sum for the all-reduce), in the places the RCCL calls take in a multi-process run.  A test harness for the shard
    arithmetic, not a way to go faster."""

    def __init__(self, engines, device):
        import torch
        self.world = len(engines)
        self.solvers = []
        for r, e in enumerate(engines):
            s = ShardedSolver(e, _LocalRank(r, self.world), device, backend="local")
            s.dist = None                       # its own exchange_* calls do nothing: the group exchanges for it
            self.solvers.append(s)
        self.torch = torch
        self.collectives = 0

    def _all_gather_sep(self):
        per = self.solvers[0].sep_per * self.solvers[0].per_rank
        for r, src in enumerate(self.solvers):
            for dst in self.solvers:
                if dst is not src:
                    dst.sep[r * per:(r + 1) * per].copy_(src.sep[r * per:(r + 1) * per])
        self.collectives += 1

    def _all_reduce(self, name):
        total = self.torch.stack([getattr(s, name) for s in self.solvers]).sum(dim=0)
        for s in self.solvers:
            getattr(s, name).copy_(total)
        self.collectives += 1

    def _solve(self):
        """one staged solve of all shards, the refinement included (ShardedSolver.trial up to the retraction)"""
        for s in self.solvers:
            s.phase_local()
        self._all_gather_sep()
        for s in self.solvers:
            s.phase_global()
        self._all_reduce("delta")
        n = self.solvers[0].eng.refine_count()
        if n:
            for s in self.solvers:
                s.eng.refine_begin()
            for _ in range(n):
                for s in self.solvers:
                    s.eng.solve_local()
                self._all_gather_sep()
                for s in self.solvers:
                    s.eng.solve_global()
                self._all_reduce("refine_delta")
                for s in self.solvers:
                    s.eng.refine_step()
            for s in self.solvers:
                s.eng.refine_end()

    def iterate(self, iterations: int):
        for s in self.solvers:
            s.begin()
        for _ in range(iterations):
            self._solve()
            for s in self.solvers:
                s.phase_finish()
        for s in self.solvers:
            s.eng.close_excursions()

    def gn_step(self, relin_threshold=1e-4):
        """one reference-compat update on every shard (ShardedSolver.gn_step)"""
        for s in self.solvers:
            s.eng.gn_begin(relin_threshold)
        self._solve()
        for s in self.solvers:
            s.eng.retract()
