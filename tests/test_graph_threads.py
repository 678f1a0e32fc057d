"""Host logic of the GraphManager surface under ThreadSanitizer: vf_graph.cpp against a device-free engine double
(tests/native/fake_engine.cpp).  Checks the two-lock discipline of GraphManager.h:103-104 / GraphManager.cpp:104-117 --
in particular that a failing vf_solve, which gives its snapshot back to the queues, never takes the graph lock while it
holds the state lock (lock-order inversion against vf_reserve_node)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_failing_solve_against_reserve_node_under_tsan(tmp_path):
    exe = tmp_path / "graph_threads"
    srcs = [os.path.join(ROOT, "tests", "native", "graph_threads.cpp"), os.path.join(ROOT, "tests", "native", "fake_engine.cpp"),
            os.path.join(ROOT, "vil_sensor_fusion_amd", "csrc", "vf_graph.cpp")]
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread", "-o", str(exe)] + srcs)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1 exitcode=66")
    p = subprocess.run([str(exe)], env=env, capture_output=True, text=True, timeout=300)
    print(p.stdout, p.stderr[-3000:])
    assert "ThreadSanitizer" not in p.stderr, "data race or lock-order inversion reported"
    assert p.returncode == 0


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_far_factor_routing_host_logic(tmp_path):
    """vf_add_between takes any pair of keys (GraphManager.cpp:83-88): band factors to vf_engine_set_between, everything else
    to the bounded far list that is re-sent through vf_engine_set_extra_between at every solve and pruned as keys leave the
    fixed-lag window (tests/native/graph_far.cpp against the engine double; the arithmetic is tests/test_gpu_far_factors.py)."""
    exe = tmp_path / "graph_far"
    srcs = [os.path.join(ROOT, "tests", "native", "graph_far.cpp"), os.path.join(ROOT, "tests", "native", "fake_engine.cpp"),
            os.path.join(ROOT, "vil_sensor_fusion_amd", "csrc", "vf_graph.cpp")]
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-pthread", "-o", str(exe)] + srcs)
    p = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    print(p.stdout, p.stderr[-2000:])
    assert p.returncode == 0 and "far-factor routing ok" in p.stdout


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_staged_graph_host_logic(tmp_path):
    """GraphManager::graph() through the C ABI (vf_graph_get_staged) against the engine double: the three priors,
    then every between factor in the order it was added -- keys, measured(), covariance --, emptied by solve()
    (GraphManager.cpp:27-35, 46-49, 83-88, 112-114; test/UnitTests.cpp:200,222-233).  No GPU needed: this is host bookkeeping."""
    exe = tmp_path / "graph_staged"
    srcs = [os.path.join(ROOT, "tests", "native", "graph_staged.cpp"), os.path.join(ROOT, "tests", "native", "fake_engine.cpp"),
            os.path.join(ROOT, "vil_sensor_fusion_amd", "csrc", "vf_graph.cpp")]
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-pthread", "-o", str(exe)] + srcs)
    p = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    print(p.stdout, p.stderr[-2000:])
    assert p.returncode == 0 and "graph_staged ok" in p.stdout
