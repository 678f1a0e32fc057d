"""CPU: the C-ABI library loads and exports every symbol include/vilfusion.h declares, and
fails loudly (no CPU fallback) when no GPU is present."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "vilfusion.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vf_[a-z0-9_]+)\s*\(", txt)) - {"vf_callback"})


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    from vil_sensor_fusion_amd import _lib
    if not os.path.exists(_lib.lib_path()):
        g.build()
    return _lib.lib()


def test_header_symbols_exported(lib):
    from vil_sensor_fusion_amd import _lib
    syms = declared_symbols()
    assert len(syms) >= 40
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/vilfusion.h but not exported"
    assert sorted(_lib.SYMBOLS) == syms


def test_version_and_error_string(lib):
    assert b"gfx950" in lib.vf_version()
    assert isinstance(lib.vf_last_error(), bytes)


def test_no_cpu_fallback_without_gpu(lib):
    n = C.c_int(-1)
    rc = lib.vf_device_count(C.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is visible here")
    from vil_sensor_fusion_amd import Engine, VilFusionError
    from vil_sensor_fusion_amd.graph_manager import GraphManager
    with pytest.raises(VilFusionError) as ei:
        Engine()
    assert ei.value.code == -7
    with pytest.raises(VilFusionError) as ei:
        GraphManager()
    assert ei.value.code == -7


def test_argument_validation_without_gpu(lib):
    assert lib.vf_engine_create(None, None) == -1
    assert lib.vf_create(None, None, None) == -1
    assert lib.vf_solve(None) == -1
    assert lib.vf_engine_iterate(None, 1) == -1


def test_product_never_imports_oracle():
    """The product package must not import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "vil_sensor_fusion_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")) or f == "Makefile":
                src = open(os.path.join(dirpath, f)).read()
                assert "vf_oracle" not in src and "from oracle" not in src and "import oracle" not in src, f
