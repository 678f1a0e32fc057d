"""The C ABI from C: examples/minimal.c compiles against include/vilfusion.h with plain gcc, links libvilfusion.so and
runs -- on a box without a GPU it must stop with VF_ERR_NO_DEVICE (exit 7: no CPU fallback), on an MI355X it must smooth
a vehicle at rest (exit 0)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    import __graft_entry__ as g
    from vil_sensor_fusion_amd import _lib
    if not os.path.exists(_lib.lib_path()):
        g.build()
    exe = str(tmp_path / "minimal")
    libdir = os.path.join(ROOT, "vil_sensor_fusion_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "minimal.c"),
                           "-L", libdir, "-lvilfusion", f"-Wl,-rpath,{libdir}", "-lm", "-o", exe])
    return exe


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_c_example_builds_and_fails_loudly_without_a_gpu(tmp_path):
    import ctypes as C
    from vil_sensor_fusion_amd import _lib
    n = C.c_int(0)
    if _lib.lib().vf_device_count(C.byref(n)) == 0 and n.value > 0:
        pytest.skip("a GPU is visible here (the gpu-marked twin of this test runs the example)")
    p = subprocess.run([_build(tmp_path)], capture_output=True, text=True, timeout=120)
    assert p.returncode == 7 and "no device" in p.stdout, (p.returncode, p.stdout, p.stderr)


@pytest.mark.gpu
def test_c_example_runs_on_the_gpu(tmp_path):
    p = subprocess.run([_build(tmp_path)], capture_output=True, text=True, timeout=300)
    print(p.stdout[-600:])
    assert p.returncode == 0, (p.returncode, p.stdout[-2000:], p.stderr[-2000:])
    assert "20 keyframes, 20 callbacks" in p.stdout


@pytest.mark.gpu
def test_c_example_built_against_the_shorter_structs_runs_on_the_gpu(tmp_path):
    """examples/minimal.c compiled against include/vilfusion.h WITHOUT the fields behind the VF_ABI_TAIL markers (a binding built
    against an older release): struct_size tells the library how much of vf_graph_opts the caller knows, the rest gets defaults"""
    from tests.test_abi import _truncated_header
    exe = str(tmp_path / "minimal_old")
    libdir = os.path.join(ROOT, "vil_sensor_fusion_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", _truncated_header(tmp_path), os.path.join(ROOT, "examples", "minimal.c"),
                           "-L", libdir, "-lvilfusion", f"-Wl,-rpath,{libdir}", "-lm", "-o", exe])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, (p.returncode, p.stdout[-2000:], p.stderr[-2000:])
    assert "20 keyframes, 20 callbacks" in p.stdout


def _build_sharded(tmp_path):
    import __graft_entry__ as g
    from vil_sensor_fusion_amd import _lib
    if not os.path.exists(_lib.lib_path()):
        g.build()
    exe = str(tmp_path / "sharded")
    libdir = os.path.join(ROOT, "vil_sensor_fusion_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "sharded.c"),
                           "-L", libdir, "-lvilfusion", "-L", "/opt/rocm/lib", "-lrccl", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-lm", "-o", exe])
    return exe


def test_shard_exchange_plan_is_what_the_python_solver_exchanges():
    """The ranges the C path (vf_shard_iterate: ncclAllGather / ncclAllReduce issued by the library) exchanges are the ones
    distributed.ShardedSolver hands to torch.distributed: rank r's contiguous slice of the chunk-major separator buffer,
    and the increments of every keyframe slot followed by one failure flag per window.  Host-only: no device needed."""
    import ctypes as C
    from vil_sensor_fusion_amd import _lib
    l = _lib.lib()
    SEPK = (2 * 27 * 28 + 27 * 27 + 7) // 8 * 8                      # vf_kernels.hpp: [sepR | sepS | sepC] of one (chunk, window), padded to 8
    for windows, capacity, chunks, world in ((1, 10008, 96, 8), (1, 10000, 96, 4), (3, 200, 6, 2), (1, 160, 4, 1)):
        M = (capacity + 63) // 64 * 64
        seen = []
        for rank in range(world):
            v = [C.c_long() for _ in range(4)]
            _lib.check(l.vf_shard_exchange_plan(windows, capacity, chunks, rank, world, *[C.byref(x) for x in v]))
            so, sc, st, dc = (x.value for x in v)
            per_rank = chunks // world
            assert sc == per_rank * windows * SEPK and so == rank * sc and st == chunks * windows * SEPK      # ShardedSolver.exchange_sep
            assert dc == windows * M * 15 + windows                                                           # ShardedSolver.delta
            seen.append((so, so + sc))
        assert seen[0][0] == 0 and seen[-1][1] == chunks * windows * SEPK and all(a[1] == b[0] for a, b in zip(seen, seen[1:]))
    assert l.vf_shard_exchange_plan(1, 100, 6, 0, 4, None, None, None, None) != 0         # 6 chunks over 4 ranks


@pytest.mark.gpu
def test_sharded_c_example_issues_its_collectives_through_rccl(tmp_path):
    """examples/sharded.c: a C program with its own one-rank RCCL communicator drives a time-sharded window through
    vf_shard_iterate (the library issues ncclAllGather / ncclAllReduce on the engine's stream) and ends with the bits of the
    unsharded engine.  N > 1 ranks stay unmeasured on this pool (one GPU)."""
    p = subprocess.run([_build_sharded(tmp_path)], capture_output=True, text=True, timeout=600)
    print(p.stdout[-1200:])
    assert p.returncode == 0, (p.returncode, p.stdout[-2000:], p.stderr[-2000:])
    assert "through RCCL" in p.stdout
