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


def test_ctypes_mirrors_match_the_header_layout(tmp_path):
    """The Python side mirrors the header's structs by hand (vil_sensor_fusion_amd/_lib.py): compile a C program against
    include/vilfusion.h that prints sizeof / offsetof of every field and compare with ctypes -- a field added to the
    header but not to the mirror (or in another place) would otherwise shift every later option silently."""
    import shutil
    import subprocess
    from vil_sensor_fusion_amd import _lib
    if shutil.which("gcc") is None:
        pytest.skip("needs gcc")
    mirrors = {"vf_engine_opts": _lib.EngineOptsC, "vf_engine_tuning": _lib.EngineTuningC, "vf_imu_params": _lib.ImuParamsC, "vf_shard_info": _lib.ShardInfoC,
               "vf_graph_opts": _lib.GraphOptsC}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "vilfusion.h"', 'int main(void) {']
    for name, cls in mirrors.items():
        lines.append(f'printf("{name} %zu\\n", sizeof({name}));')
        for f, _ in cls._fields_:
            lines.append(f'printf("{name}.{f} %zu\\n", offsetof({name}, {f}));')
    lines += ["return 0; }"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).splitlines())
    for name, cls in mirrors.items():
        assert int(got[name]) == C.sizeof(cls), name
        for f, _ in cls._fields_:
            assert int(got[f"{name}.{f}"]) == getattr(cls, f).offset, f"{name}.{f}"
    # and the header declares no field the mirror lacks (count the members of each typedef'd struct)
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "vilfusion.h")).read(), flags=re.S)
    for name, cls in mirrors.items():
        body = re.search(r"typedef struct \{([^}]*)\} " + name + ";", txt, flags=re.S).group(1)
        members = [m for decl in body.split(";") if decl.strip() for m in decl.split(",")]
        assert len(members) == len(cls._fields_), (name, len(members), len(cls._fields_))


def test_defaults_do_not_depend_on_the_environment(lib, monkeypatch):
    """Solver-form switches are option fields: the defaults the library hands out are the same whatever
    the caller's environment says, and the sources read no VF_* variable that changes what is computed."""
    from vil_sensor_fusion_amd import _lib
    outs = []
    for env in ({}, {"VF_FUSED": "1", "VF_HYBRID_T": "7", "VF_NO_HYBRID": "1", "VF_USE_GRAPH": "1", "VF_NO_WARM": "1",
                     "VF_TWISTED_MAX_WINDOWS": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        o, t = _lib.EngineOptsC(), _lib.EngineTuningC()
        lib.vf_engine_default_opts(C.byref(o))
        lib.vf_engine_default_tuning(C.byref(t))
        assert o.struct_size == C.sizeof(o) and t.struct_size == C.sizeof(t)
        outs.append((o.chunks, t.sweep_two_sided_max, t.hybrid_threshold, o.cold_start, t.use_hip_graph, t.solve_split_min, t.solve_assemble_min))
    assert outs[0] == outs[1] == (0, 256, 256, 0, 0, 2048, 768)
    csrc = os.path.join(ROOT, "vil_sensor_fusion_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".cpp", ".hpp")):
            for m in re.findall(r'getenv\("([A-Z_]+)"\)', open(os.path.join(csrc, f)).read()):
                assert m == "VF_SOLVE_TIMING", (f, m)        # (prints lap times of vf_solve to stderr; changes nothing computed)


def _truncated_header(tmp_path):
    """include/vilfusion.h as it was before the fields behind the VF_ABI_TAIL markers were added: what a binding compiled
    against an older release of the header holds"""
    txt = open(os.path.join(ROOT, "include", "vilfusion.h")).read()
    cut = re.sub(r"/\* VF_ABI_TAIL.*?\*/.*?(\} vf_(engine|graph)_opts;)", r"\1", txt, flags=re.S)
    assert cut.count("int incremental;") == 0 and txt.count("int incremental;") == 2
    d = tmp_path / "old_include"
    d.mkdir()
    (d / "vilfusion.h").write_text(cut)
    return str(d)


def test_a_caller_built_against_a_shorter_struct_is_served(lib, tmp_path):
    """vf_engine_opts / vf_graph_opts carry struct_size.  A C caller compiled against the header WITHOUT the newest
    fields gets its (shorter) struct filled with defaults -- not one byte beyond it written -- and its create call is accepted
    (the library supplies the defaults of what it does not know); a size the library does not know is refused.  No GPU needed:
    the size checks come before the device is touched (the gpu-marked twin runs examples/minimal.c built this way)."""
    import shutil
    import subprocess
    from vil_sensor_fusion_amd import _lib
    if shutil.which("gcc") is None:
        pytest.skip("needs gcc")
    src = tmp_path / "old_caller.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "vilfusion.h"
int main(void) {
    struct { vf_engine_opts o; unsigned char guard[64]; } a;
    struct { vf_graph_opts o; unsigned char guard[64]; } b;
    memset(&a, 0xAB, sizeof(a));
    memset(&b, 0xAB, sizeof(b));
    vf_engine_default_opts(&a.o);
    vf_graph_default_opts(&b.o);
    for (int i = 0; i < 64; i++) if (a.guard[i] != 0xAB || b.guard[i] != 0xAB) { printf("overrun\n"); return 2; }
    printf("engine %u %zu graph %u %zu\n", a.o.struct_size, sizeof(a.o), b.o.struct_size, sizeof(b.o));
    if (a.o.struct_size != sizeof(a.o) || b.o.struct_size != sizeof(b.o)) return 3;
    if (a.o.windows != 1 || a.o.gauge_floor != 3e-4 || b.o.relin_threshold != 1e-4 || b.o.iterations != 5) return 4;
    vf_engine* e = 0;
    a.o.struct_size = 4096;                       /* a caller from the future */
    int rc = vf_engine_create(&a.o, &e);
    printf("future caller: %d %s\n", rc, vf_last_error());
    if (rc != VF_ERR_INVALID) return 5;
    a.o.struct_size = 0;                          /* a caller that never filled the struct */
    if (vf_engine_create(&a.o, &e) != VF_ERR_INVALID) return 6;
    a.o.struct_size = (unsigned)sizeof(a.o);
    rc = vf_engine_create(&a.o, &e);              /* accepted as far as the struct goes: fails (or not) on the device only */
    printf("old caller: %d %s\n", rc, rc ? vf_last_error() : "ok");
    if (rc == VF_ERR_INVALID) return 7;
    if (rc == 0) vf_engine_destroy(e);
    return 0;
}
''')
    exe = tmp_path / "old_caller"
    libdir = os.path.join(ROOT, "vil_sensor_fusion_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", _truncated_header(tmp_path), str(src), "-L", libdir, "-lvilfusion",
                           f"-Wl,-rpath,{libdir}", "-o", str(exe)])
    p = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    print(p.stdout)
    assert p.returncode == 0, (p.returncode, p.stdout, p.stderr)
    sizes = p.stdout.splitlines()[0].split()
    assert int(sizes[1]) < C.sizeof(_lib.EngineOptsC) and int(sizes[4]) < C.sizeof(_lib.GraphOptsC)      # really the shorter structs
