#!/usr/bin/env python3
"""Build differently compiled copies of the library for tools/ab_variants.sh (development aid; build container).

usage: python tools/build_variants.py name=[main|w2|both]:flag,flag,... [name=...]      ("base=main:" = the shipped flags)
The flags replace / extend those of ONE translation unit (main = orr_kernels.hip: everything incl. the one-wave step kernel, ILP
scheduler; w2 = orr_kernels_w2.hip: the two-waves-per-SIMD step kernel, default scheduler); the other unit keeps its shipped flags.
A flag set that names an -amdgpu-sched-strategy replaces the unit's own; "nosched" removes it; -O1/-O2/-O3 replace -O2.
Example: python tools/build_variants.py base=main: w2ilp=w2:-mllvm,-amdgpu-sched-strategy=iterative-ilp w2o3=w2:-O3
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openroborl_amd import _lib  # noqa: E402


def apply(base, extra):
    base = [f for f in base if f != "-shared"]
    extra = list(extra)

    def drop_sched(fl):
        for k, f in enumerate(fl):
            if "amdgpu-sched-strategy" in f:
                del fl[k - 1:k + 1]
                return
    if any("amdgpu-sched-strategy" in f for f in extra) or "nosched" in extra:
        drop_sched(base)
    if "nosched" in extra:
        extra.remove("nosched")
    if any(f in ("-O1", "-O2", "-O3", "-Os") for f in extra):
        base = [f for f in base if f != "-O2"]
    return base + extra


for spec in sys.argv[1:]:
    name, _, rest = spec.partition("=")
    tu, _, fl = rest.partition(":")
    extra = [f for f in fl.split(",") if f]
    fm = apply(_lib.HIPCC_FLAGS, extra if tu in ("main", "both") else [])
    fw = apply(_lib.HIPCC_FLAGS_W2, extra if tu in ("w2", "both") else [])
    out = os.path.join(ROOT, "openroborl_amd", "lib_var_%s.so" % name)
    tmp = "/tmp/var_%s" % name
    os.makedirs(tmp, exist_ok=True)
    procs = [subprocess.Popen([_lib.HIPCC] + fm + ["-c", '-DORR_SOURCE_HASH="variant-%s"' % name, "-o", tmp + "/k.o", _lib.SRC]),
             subprocess.Popen([_lib.HIPCC] + fw + ["-c", "-o", tmp + "/w.o", _lib.SRC_W2]),
             subprocess.Popen([_lib.HIPCC, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-c", "-o", tmp + "/p.o", _lib.SRC_POLICY]),
             subprocess.Popen([_lib.HIPCC, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-c", "-o", tmp + "/l.o", _lib.SRC_LEARNER])]
    assert all(p.wait() == 0 for p in procs), spec
    subprocess.check_call([_lib.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, tmp + "/k.o", tmp + "/w.o", tmp + "/p.o", tmp + "/l.o"])
    print("built", out, tu, " ".join(extra))
