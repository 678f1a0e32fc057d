"""One process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" in CPU tests).

The hot path shards over independent smoothing windows (vehicles / sequences): every rank owns a
contiguous block of windows and runs the whole K0-K5 pipeline on its own MI355X; there is NO
data-path collective.  The only communication is control-plane: a barrier around the timed region,
a MAX-reduce of the elapsed time and an all-gather of per-rank summaries.  (Sharding ONE window in
time with an RCCL reduce of the separator system is the next multi-GPU row, DESIGN.md section e.)
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
