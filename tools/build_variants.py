#!/usr/bin/env python3
"""Build differently compiled copies of the library for tools/ab_variants.sh (development aid; build container).

usage: python tools/build_variants.py name=flag,flag,... [name=...]      ("base" = the shipped flags)
Each variant replaces the -mllvm -amdgpu-sched-strategy=... pair of the shipped flags when it names one of its own, and adds the rest.
Example: python tools/build_variants.py base= minreg=-mllvm,-amdgpu-sched-strategy=iterative-minreg o3=-O3
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openroborl_amd import _lib  # noqa: E402

for spec in sys.argv[1:]:
    name, _, fl = spec.partition("=")
    extra = [f for f in fl.split(",") if f]
    base = [f for f in _lib.HIPCC_FLAGS if f != "-shared"]
    if any("amdgpu-sched-strategy" in f for f in extra):
        i = next(k for k, f in enumerate(base) if "amdgpu-sched-strategy" in f)
        del base[i - 1:i + 1]
    if any(f in ("-O1", "-O2", "-O3", "-Os") for f in extra):
        base = [f for f in base if f not in ("-O2",)]
    if "nosched" in extra:
        extra.remove("nosched")
        i = next(k for k, f in enumerate(base) if "amdgpu-sched-strategy" in f)
        del base[i - 1:i + 1]
    out = os.path.join(ROOT, "openroborl_amd", "lib_var_%s.so" % name)
    tmp = "/tmp/var_%s" % name
    os.makedirs(tmp, exist_ok=True)
    subprocess.check_call([_lib.HIPCC] + base + extra + ["-c", '-DORR_SOURCE_HASH="variant-%s"' % name, "-o", tmp + "/k.o", _lib.SRC])
    subprocess.check_call([_lib.HIPCC, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-c", "-o", tmp + "/p.o", _lib.SRC_POLICY])
    subprocess.check_call([_lib.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, tmp + "/k.o", tmp + "/p.o"])
    print("built", out, " ".join(extra))
