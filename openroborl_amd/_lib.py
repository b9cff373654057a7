"""Build + load the C-ABI library (libopenroborl_hip.so) and declare its ctypes signatures.

There is NO CPU fallback: if the library cannot be built or loaded, importing / using the
environment raises.  (The CPU restatement under oracle/ is test infrastructure and is never
imported from here.)
"""
import ctypes as C
import os
import subprocess

from . import _abi

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(PKG_DIR, "csrc", "orr_kernels.hip")
SRC_POLICY = os.path.join(PKG_DIR, "csrc", "orr_policy.hip")
DEPS = [SRC, SRC_POLICY] + [os.path.join(PKG_DIR, "csrc", h) for h in ("orr_device.h", "orr_robot_io.h", "orr_physics.h", "orr_task.h")] + [
        os.path.join(os.path.dirname(PKG_DIR), "include", "openroborl_hip.h"),
        os.path.join(os.path.dirname(PKG_DIR), "include", "openroborl_policy.h")]
LIB_PATH = os.path.join(PKG_DIR, "libopenroborl_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# The step kernel runs one wave per SIMD with up to 512 VGPRs, so a wave's issued instruction count is what bounds it:
# schedule for instruction-level parallelism rather than occupancy, and keep the SLP vectoriser off (its packed-fp32
# operations cost more register moves than they save).  Measured on MI355X, 4096 robots: -O3 default 7.1 M env steps/s,
# + iterative-ilp 7.5 M, + no SLP 7.9 M, -O2 8.1 M.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O2", "-fPIC", "-shared", "-std=c++17", "-Wno-unused-value", "-fno-slp-vectorize",
               "-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]

EXPORTS = [
    "orr_last_error", "orr_abi_version", "orr_state_stride", "orr_layout_count", "orr_layout_name",
    "orr_layout_offset", "orr_layout_size", "orr_layout_is_int", "orr_sizeof_config", "orr_sizeof_model",
    "orr_create", "orr_destroy", "orr_set_model", "orr_set_motion", "orr_bind", "orr_reset", "orr_step",
    "orr_time_steps",
    "orr_policy_packed_size", "orr_policy_pack", "orr_policy_forward", "orr_gae",
]


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(d) > t for d in DEPS if os.path.exists(d))


def build(force=False, verbose=False, out_path=None, extra_flags=()):
    """Compile the HIP kernels + C-ABI for gfx950 into the in-tree shared library (or `out_path`, with `extra_flags`
    for the env kernels: development builds such as tools/phase_cycles.py)."""
    if out_path is None:
        if not force and not needs_build():
            return LIB_PATH
        out_path = LIB_PATH
    flags = [f for f in HIPCC_FLAGS if f != "-shared"] + ["-c"]
    for var in ("ORR_WAVES_PER_EU", "ORR_LANES_PER_ROBOT"):      # tuning experiments only
        if os.environ.get(var):
            flags.append("-D%s=%d" % (var, int(os.environ[var])))
    for d in os.environ.get("ORR_EXTRA_DEFS", "").split():
        flags.append("-D" + d)
    flags += list(extra_flags)
    tag = "" if out_path == LIB_PATH else "." + os.path.basename(out_path)
    obj_env = os.path.join(PKG_DIR, "csrc", "orr_kernels%s.o" % tag)
    obj_pol = os.path.join(PKG_DIR, "csrc", "orr_policy%s.o" % tag)
    cmds = [[HIPCC] + flags + ["-o", obj_env, SRC],
            # the policy forward pass (matrix cores) is its own translation unit with the compiler's default scheduling
            [HIPCC, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-c", "-o", obj_pol, SRC_POLICY],
            [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out_path, obj_env, obj_pol]]
    for cmd in cmds:
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    if tag:   # development builds leave no objects behind
        for o in (obj_env, obj_pol):
            if os.path.exists(o):
                os.remove(o)
    return out_path


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if needs_build() and not os.environ.get("ORR_LIB_PATH"):
        try:
            build()
        except Exception as e:  # stale library on a box without hipcc is still usable
            if not os.path.exists(LIB_PATH):
                raise RuntimeError("libopenroborl_hip.so is missing and could not be built: %r" % (e,))
    L = C.CDLL(os.environ.get("ORR_LIB_PATH", LIB_PATH))   # override = tuning experiments (A/B of two builds)
    vp = C.c_void_p
    L.orr_last_error.restype = C.c_char_p
    L.orr_abi_version.restype = C.c_int32
    L.orr_create.restype = C.c_int32
    L.orr_create.argtypes = [C.POINTER(_abi.OrrConfig), C.POINTER(vp)]
    L.orr_destroy.argtypes = [vp]
    L.orr_set_model.restype = C.c_int32
    L.orr_set_model.argtypes = [vp, C.c_int32, C.POINTER(_abi.OrrModel)]
    L.orr_set_motion.restype = C.c_int32
    L.orr_set_motion.argtypes = [vp, C.c_int32, vp, vp, C.c_int32, C.c_float, C.c_int32, C.POINTER(C.c_float)]
    L.orr_bind.restype = C.c_int32
    L.orr_bind.argtypes = [vp, vp, vp, vp, C.c_int32]
    L.orr_reset.restype = C.c_int32
    L.orr_reset.argtypes = [vp, vp, vp, vp]
    L.orr_step.restype = C.c_int32
    L.orr_step.argtypes = [vp, vp, vp, vp, vp, vp]
    L.orr_debug_physics.restype = C.c_int32
    L.orr_debug_physics.argtypes = [vp, vp, vp, C.c_int32, vp]
    L.orr_time_steps.restype = C.c_int32
    L.orr_time_steps.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int32, C.POINTER(C.c_float)]
    L.orr_sizeof_config.restype = C.c_int32
    L.orr_sizeof_model.restype = C.c_int32
    # include/openroborl_policy.h
    L.orr_policy_packed_size.restype = C.c_int64
    L.orr_policy_packed_size.argtypes = [C.c_int32, C.c_int32]
    L.orr_policy_pack.restype = C.c_int32
    L.orr_policy_pack.argtypes = [vp, C.c_int32, C.c_int32, vp, vp]
    L.orr_gae.restype = C.c_int32
    L.orr_gae.argtypes = [vp, vp, vp, vp, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_int32, C.c_float, vp, vp, vp]
    L.orr_policy_forward.restype = C.c_int32
    L.orr_policy_forward.argtypes = [C.POINTER(_abi.OrrPolicyNet), vp, C.c_int32, vp, C.c_float, C.c_float, vp, vp, vp, vp, vp]
    if L.orr_abi_version() != _abi.ABI_VERSION:
        raise RuntimeError("libopenroborl_hip.so ABI version mismatch")
    if L.orr_sizeof_config() != C.sizeof(_abi.OrrConfig) or L.orr_sizeof_model() != C.sizeof(_abi.OrrModel):
        raise RuntimeError("ctypes struct layout does not match include/openroborl_hip.h")
    _lib = L
    return L


def check(rc, lib=None):
    if rc != 0:
        lib = lib or load()
        raise RuntimeError("openroborl_hip error %d: %s" % (rc, lib.orr_last_error().decode()))
