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
