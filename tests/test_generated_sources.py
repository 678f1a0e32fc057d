"""The panel factorisation of K4 is generated code (tools/gen_pivot.py -> csrc/vf_pivot_*.inc).  The committed files
must be what the generator produces, and must respect the ordering rules the generator's docstring states."""
import importlib.util
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gen():
    spec = importlib.util.spec_from_file_location("gen_pivot", os.path.join(ROOT, "tools", "gen_pivot.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_committed_pivot_code_is_what_the_generator_writes():
    g = _gen()
    for n, kw in ((15, dict(near=3)), (27, dict(near=27))):
        path = os.path.join(ROOT, "vil_sensor_fusion_amd", "csrc", f"vf_pivot_{n}.inc")
        assert open(path).read() == g.gen(n, **kw), path
    path = os.path.join(ROOT, "vil_sensor_fusion_amd", "csrc", "vf_pivot_15p.inc")        # the lane-swap variant
    assert open(path).read() == g.gen(15, near=3, rep="permlane")
    assert "S[pv_bcw]" not in open(path).read() and open(path).read().count("rep_row0(") == 11


def test_column_updates_arrive_before_the_column_is_a_pivot():
    """Every column c2 receives exactly one update from each pivot c < c2, and all of them are issued before the statement
    that reads its diagonal entry (the v_readlane of the next pivot) -- for the v_readlane and the DPP form alike."""
    g = _gen()
    for n, kw in ((15, dict(near=3)), (27, dict(near=27))):
        seen = {c2: set() for c2 in range(n)}
        for line in g.gen(n, **kw).splitlines():
            m = re.search(r"p\[(\d+)\] = fma\(-p\[(\d+)\]", line)
            if m:
                seen[int(m.group(1))].add(int(m.group(2)))
            m = re.search(r'row_newbcast:(\d+) .*"\+v"\(p\[(\d+)\]\) : "v"\(pv_b(\d+)\)', line)
            if m:
                assert m.group(1) == m.group(2)          # multiplier lane = the column's own pivot-block row
                seen[int(m.group(2))].add(int(m.group(3)))
            m = re.search(r"pv_d = readlane_d\(p\[(\d+)\], (\d+)\);", line)
            if m:
                c = int(m.group(1))
                assert m.group(1) == m.group(2) and seen[c] == set(range(c)), (n, c, sorted(seen[c]))
        for c2 in range(n):
            assert seen[c2] == set(range(c2)), (n, c2)


def test_the_librarys_kernels_are_the_committed_list():
    """vf_kernels.hip is one translation unit cut into section files (csrc/kernels/*.inc): moving code between them must not
    add, drop or rename a kernel.  csrc/kernels.list is the committed list (host stubs of libvilfusion.so, one per kernel
    template instance); a deliberate change regenerates it with the command in the assertion message."""
    import re
    import subprocess
    import __graft_entry__ as g
    from vil_sensor_fusion_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(_lib.lib_path()):
        g.build()
    out = subprocess.check_output(["nm", "-C", _lib.lib_path()], text=True)
    names = sorted(set(re.sub(r"\(.*", "", l.split("__device_stub__")[1]).strip() for l in out.splitlines() if "__device_stub__" in l))
    want = open(os.path.join(root, "vil_sensor_fusion_amd", "csrc", "kernels.list")).read().split("\n")[:-1]
    assert names == want, ("kernels of libvilfusion.so differ from csrc/kernels.list (after a deliberate change: python tools/gen_kernel_list.py)",
                           sorted(set(names) ^ set(want)))
    for sec in os.listdir(os.path.join(root, "vil_sensor_fusion_amd", "csrc", "kernels")):
        assert sec.endswith(".inc") and f'#include "kernels/{sec}"' in open(os.path.join(root, "vil_sensor_fusion_amd", "csrc", "vf_kernels.hip")).read(), sec
