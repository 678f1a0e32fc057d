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


def test_bench_spawns_one_child_per_gpu(tmp_path):
    """bench.py --gpus N without WORLD_SIZE starts N fresh ranks with the torch.distributed environment set, forwards the
    arguments and returns a non-zero status if any rank fails (no GPU needed: the children here are a stub script)."""
    import argparse
    import subprocess  # noqa: F401
    sys.path.insert(0, ROOT)
    import bench
    stub = tmp_path / "rank_stub.py"
    stub.write_text("import os, sys\n"
                    "open(os.path.join(sys.argv[1], 'rank%s' % os.environ['RANK']), 'w').write(' '.join(\n"
                    "    [os.environ[k] for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR')] + sys.argv[2:]))\n"
                    "sys.exit(3 if os.environ['RANK'] == sys.argv[2] else 0)\n")
    rc = bench.spawn_ranks(argparse.Namespace(gpus=3), script=str(stub), argv=[str(tmp_path), "none", "--steps", "5"])
    assert rc == 0
    got = sorted((tmp_path / f"rank{r}").read_text() for r in range(3))
    assert got == [f"{r} {r} 3 127.0.0.1 none --steps 5" for r in range(3)]
    assert bench.spawn_ranks(argparse.Namespace(gpus=2), script=str(stub), argv=[str(tmp_path), "1"]) == 3


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=120)
    assert res.returncode != 0 and "WORLD_SIZE=1" in res.stderr
