#!/usr/bin/env python3
"""Regenerate vil_sensor_fusion_amd/csrc/kernels.list from the built library (the host stubs of libvilfusion.so, one per kernel
template instance).  tests/test_generated_sources.py compares the library with the committed list: run this after a
DELIBERATE change of the kernel set."""
import os
import re
import subprocess

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.check_output(["nm", "-C", os.path.join(root, "vil_sensor_fusion_amd", "libvilfusion.so")], text=True)
names = sorted(set(re.sub(r"\(.*", "", l.split("__device_stub__")[1]).strip() for l in out.splitlines() if "__device_stub__" in l))
open(os.path.join(root, "vil_sensor_fusion_amd", "csrc", "kernels.list"), "w").write("\n".join(names) + "\n")
print(len(names), "kernels")
