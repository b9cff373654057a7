#!/usr/bin/env python3
"""Search for ORR_PARITY (openroborl_amd/csrc/orr_device.h: start parity of the hand-written Gauss-Seidel loops; development aid).

usage:  here (build container):  python tools/parity_search.py build <mask> [<mask> ...]   -> openroborl_amd/lib_var_par<mask>.so
        on the GPU box (gpurun):   bash tools/ab_variants.sh                                -> env steps/s and kernel ms per variant
The masks to try next are chosen by hand from the previous round's result (one bit at a time, keep it if the kernel got faster)."""
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openroborl_amd import _lib  # noqa: E402

if sys.argv[1] == "build":
    base = [f for f in _lib.HIPCC_FLAGS if f != "-shared"]
    procs = []
    for m in sys.argv[2:]:
        mask = int(m, 0)
        name = "par%04x" % mask
        procs.append((name, subprocess.Popen([_lib.HIPCC] + base + ["-DORR_PARITY=0x%x" % mask, '-DORR_SOURCE_HASH="x"', "-c", "-o", "/tmp/%s_k.o" % name, _lib.SRC],
                                             stderr=subprocess.PIPE, text=True)))
    subprocess.check_call([_lib.HIPCC, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-c", "-o", "/tmp/par_p.o", _lib.SRC_POLICY])
    for name, p in procs:
        p.wait()
        if p.returncode:
            print(name, "FAILED", p.stderr.read()[-400:])
            continue
        subprocess.check_call([_lib.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", "%s/openroborl_amd/lib_var_%s.so" % (ROOT, name), "/tmp/%s_k.o" % name, "/tmp/par_p.o"])
        print("built", name)
