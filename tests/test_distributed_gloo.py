"""CPU, world_size 2, gloo: the N>1 control plane of bench.py (window sharding, barrier,
MAX-over-ranks timing, summary gather).  There is no data-path collective to test: windows are
independent (DESIGN.md section e)."""
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total_windows, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from vil_sensor_fusion_amd import distributed as D
    dist = D.init(backend="gloo")
    lo, hi = D.shard_windows(total_windows, rank, world)
    D.barrier(dist)
    elapsed = 0.5 + 0.25 * rank                      # pretend rank 1 is slower
    t = D.max_over_ranks(dist, elapsed)
    steps = 3
    summ = D.gather_summaries(dist, dict(rank=rank, windows=(lo, hi), keyframes=(hi - lo) * steps))
    value = D.whole_job_throughput(summ, t)
    D.barrier(dist)
    q.put((rank, lo, hi, t, value, [s["windows"] for s in summ]))
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [7, 1024])
def test_two_rank_control_plane(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, t0, v0, w0), (r1, lo1, hi1, t1, v1, w1) = res
    assert (lo0, hi1) == (0, total) and hi0 == lo1            # disjoint cover, contiguous
    assert abs((hi0 - lo0) - (hi1 - lo1)) <= 1                # balanced
    assert t0 == t1 == 0.75                                   # MAX over ranks
    assert v0 == v1 == total * 3 / 0.75                       # whole-job aggregate
    assert w0 == w1 == [(lo0, hi0), (lo1, hi1)]


def test_shard_windows_properties():
    sys.path.insert(0, ROOT)
    from vil_sensor_fusion_amd.distributed import shard_windows
    for total in (0, 1, 5, 8, 1000):
        for world in (1, 2, 3, 8):
            spans = [shard_windows(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
