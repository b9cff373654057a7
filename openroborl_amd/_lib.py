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
SRC_ANCHOR = os.path.join(PKG_DIR, "csrc", "orr_kernels_anchor.hip")   # the friction-anchor variants of the step kernel (ABI v5): their own translation unit
SRC_W2 = os.path.join(PKG_DIR, "csrc", "orr_kernels_w2.hip")      # the two-waves-per-SIMD step kernel: its own translation unit + flags
SRC_POLICY = os.path.join(PKG_DIR, "csrc", "orr_policy.hip")
SRC_LEARNER = os.path.join(PKG_DIR, "csrc", "orr_learner.hip")    # the non-GEMM part of the PPO update (include/openroborl_learner.h)
DEPS = [SRC, SRC_W2, SRC_ANCHOR, SRC_POLICY, SRC_LEARNER] + [os.path.join(PKG_DIR, "csrc", h) for h in ("orr_device.h", "orr_robot_io.h", "orr_physics.h", "orr_task.h")] + [
        os.path.join(os.path.dirname(PKG_DIR), "include", "openroborl_hip.h"),
        os.path.join(os.path.dirname(PKG_DIR), "include", "openroborl_policy.h"),
        os.path.join(os.path.dirname(PKG_DIR), "include", "openroborl_learner.h")]
LIB_PATH = os.path.join(PKG_DIR, "libopenroborl_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# The step kernel runs one wave per SIMD with up to 512 VGPRs, so a wave's issued instruction count is what bounds it:
# schedule for instruction-level parallelism rather than occupancy, and keep the SLP vectoriser off (its packed-fp32
# operations cost more register moves than they save).  Measured on MI355X, 4096 robots: -O3 default 7.1 M env steps/s,
# + iterative-ilp 7.5 M, + no SLP 7.9 M, -O2 8.1 M.
# -amdgpu-atomic-optimizer-strategy=None: the optimizer merges the atomics of a wave into one and hands every lane its share with
# v_readfirstlane right after it, i.e. it waits for the L2 round trip on the spot; the episode-log slot atomic is issued before
# the reset and consumed after it instead, and the per-launch counters are combined per wave by hand (orr_step_kernel).
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O2", "-fPIC", "-shared", "-std=c++17", "-Wno-unused-value", "-fno-slp-vectorize",
               "-mllvm", "-amdgpu-sched-strategy=iterative-ilp", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]

# The second translation unit (orr_kernels_w2.hip = the step kernel compiled for two waves per SIMD, used for batches of more than
# 4 x #SIMDs robots) used the compiler's DEFAULT scheduler up to round 3: at 256 registers that variant spills, and the ILP schedule's long live
# ranges cost it 8 % (8192 robots: 0.382 -> 0.352 ms), while the default scheduler costs the one-wave variant 9 % (tools/ab_variants.sh).
# -O3 instead of -O2: another 1.5 % for this unit (0.3554 -> 0.3500 ms; the schedule-metric bias, the AMDGPU pressure trackers and the
# high-pressure reschedule stage make no difference, the SLP vectoriser costs 70 %).
# Round 4: of the compiler's other scheduling strategies the register-pressure-minded `iterative-maxocc` suits this unit best
# (8192 robots, interleaved A/B: default 0.3086 ms, iterative-maxocc 0.3022, max-memory-clause 0.3090, iterative-minreg 0.3207; the one-wave
# unit keeps iterative-ilp: maxocc 0.2218 against 0.2180, max-ilp 0.2349, max-memory-clause 0.2329; profiles/r04_ab11_8192.txt, r04_ab12_4096.txt).
# Round 4, last sweep on the final code (tools/build_variants.py, four interleaved rounds, profiles/r04_ab25..29_8192.log): -Os instead of -O3
# together with the priority knobs of the kernel (turns of four sub-steps, high priority past the loop) 0.2988 -> 0.2976 ms; -Os alone
# is within the noise (0.2987), -Oz costs 1 %.  (What -Os changes here is the register allocation's luck, not the amount of code: every
# function of the kernel is forced inline and every loop unrolled by pragma.)
HIPCC_FLAGS_W2 = ["-Os" if f == "-O2" else ("-amdgpu-sched-strategy=iterative-maxocc" if "amdgpu-sched-strategy" in f else f) for f in HIPCC_FLAGS]

EXPORTS = [
    "orr_last_error", "orr_abi_version", "orr_source_hash", "orr_state_stride", "orr_layout_count", "orr_layout_name",
    "orr_layout_offset", "orr_layout_size", "orr_layout_is_int", "orr_sizeof_config", "orr_sizeof_model",
    "orr_create", "orr_destroy", "orr_set_seed", "orr_set_model", "orr_set_motion", "orr_bind", "orr_reset", "orr_step",
    "orr_episode_stats", "orr_time_steps", "orr_stress_actions", "orr_debug_physics", "orr_debug_replay_step", "orr_debug_replay_reset",
    "orr_policy_packed_size", "orr_policy_pack", "orr_policy_forward", "orr_gae", "orr_gae_flags",
    "orr_learner_workspace_floats", "orr_ppo_head", "orr_relu_backward", "orr_head_backward", "orr_colsum_finish", "orr_learner_partial_rows", "orr_adam_step",
]


HASH_MARKER = b"ORR_SRC_HASH="      # the library embeds "ORR_SRC_HASH=<32 hex>" (orr_kernels.hip); read WITHOUT dlopen, see library_hash


def tuning_defines():
    """-D flags taken from the environment (tuning experiments).  They are part of the hash: a library built with them is not the
    shipped library, and setting them makes the in-tree library stale."""
    defs = []
    for var in ("ORR_WAVES_PER_EU", "ORR_LANES_PER_ROBOT"):
        if os.environ.get(var):
            defs.append("-D%s=%d" % (var, int(os.environ[var])))
    for d in os.environ.get("ORR_EXTRA_DEFS", "").split():
        defs.append("-D" + d)
    return defs


def source_hash(extra_flags=()):
    """sha256 over the sources the library is built from (file names + contents), the compiler flags and every -D the build adds."""
    import hashlib
    h = hashlib.sha256()
    for d in sorted(DEPS):
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    h.update(" ".join(HIPCC_FLAGS).encode())
    h.update(" ".join(HIPCC_FLAGS_W2).encode())
    extra = tuning_defines() + list(extra_flags)
    if extra:
        h.update(("|" + " ".join(extra)).encode())
    return h.hexdigest()[:32]


def library_hash(path=None):
    """The source hash embedded in a built library, or None.  The file is scanned for the marker, NOT dlopen'ed: ctypes never
    dlcloses, and glibc hands a later CDLL(path) of a replaced file the handle of the old mapping, so a check that loads the
    stale library would make this process run it after the rebuild."""
    path = path or LIB_PATH
    try:
        with open(path, "rb") as f:
            blob = f.read()
    except OSError:
        return None
    i = blob.find(HASH_MARKER)
    if i < 0:
        return None
    j = i + len(HASH_MARKER)
    digest = blob[j:j + 32]
    if len(digest) != 32 or any(c not in b"0123456789abcdef" for c in digest):
        return None
    return digest.decode()


def needs_build():
    """True when the in-tree library is missing or was not built from the sources on disk (content hash, not mtimes:
    a snapshot copied to another box keeps its .so but not necessarily its timestamps)."""
    return library_hash() != source_hash()


def build(force=False, verbose=False, out_path=None, extra_flags=()):
    """Compile the HIP kernels + C-ABI for gfx950 into the in-tree shared library (or `out_path`, with `extra_flags`
    for the env kernels: development builds such as tools/phase_cycles.py).  Concurrent callers (one per rank under
    torchrun) are serialised by a file lock; objects and the library are written under process-unique names and the
    library is moved into place atomically, so nobody can dlopen a half-written file."""
    import fcntl
    dev_build = out_path is not None
    if out_path is None:
        out_path = LIB_PATH
    with open(os.path.join(PKG_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not dev_build and not force and not needs_build():
                return LIB_PATH              # another rank built it while this one waited for the lock
            flags = [f for f in HIPCC_FLAGS if f != "-shared"] + ["-c", '-DORR_SOURCE_HASH="%s"' % source_hash(extra_flags)]
            flags += tuning_defines() + list(extra_flags)
            flags_w2 = [f for f in HIPCC_FLAGS_W2 if f != "-shared"] + ["-c"] + tuning_defines() + list(extra_flags)
            tag = ".%d" % os.getpid()
            obj_env = os.path.join(PKG_DIR, "csrc", "orr_kernels%s.o" % tag)
            obj_w2 = os.path.join(PKG_DIR, "csrc", "orr_kernels_w2%s.o" % tag)
            obj_an = os.path.join(PKG_DIR, "csrc", "orr_kernels_anchor%s.o" % tag)
            obj_pol = os.path.join(PKG_DIR, "csrc", "orr_policy%s.o" % tag)
            obj_lrn = os.path.join(PKG_DIR, "csrc", "orr_learner%s.o" % tag)
            tmp_so = out_path + tag + ".tmp"
            cmds = [[HIPCC] + flags + ["-o", obj_env, SRC],
                    [HIPCC] + flags_w2 + ["-o", obj_w2, SRC_W2],
                    # the friction-anchor variants of the step kernel (optional physics feature): the main unit's flags, their own unit
                    [HIPCC] + [f for f in flags if not f.startswith("-DORR_SOURCE_HASH")] + ["-o", obj_an, SRC_ANCHOR],
                    # the policy forward pass (matrix cores) is its own translation unit with the compiler's default scheduling
                    [HIPCC, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-c", "-o", obj_pol, SRC_POLICY],
                    [HIPCC, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-c", "-o", obj_lrn, SRC_LEARNER],
                    [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp_so, obj_env, obj_w2, obj_an, obj_pol, obj_lrn]]
            try:
                procs = []
                for cmd in cmds[:-1]:           # the compiles are independent: run them side by side
                    if verbose:
                        print(" ".join(cmd))
                    procs.append(subprocess.Popen(cmd))
                rcs = [p.wait() for p in procs]
                if any(rcs):
                    raise subprocess.CalledProcessError(next(r for r in rcs if r), cmds[rcs.index(next(r for r in rcs if r))])
                if verbose:
                    print(" ".join(cmds[-1]))
                subprocess.check_call(cmds[-1])
                os.replace(tmp_so, out_path)
            finally:
                for o in (obj_env, obj_w2, obj_an, obj_pol, obj_lrn, tmp_so):
                    if os.path.exists(o):
                        os.remove(o)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return out_path


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("ORR_LIB_PATH", LIB_PATH)   # override = tuning experiments (A/B of two builds)
    if path == LIB_PATH and needs_build():
        try:
            build()
        except Exception as e:
            # never run kernels that do not match the sources silently: a stale library is an error unless explicitly allowed
            if os.path.exists(LIB_PATH) and os.environ.get("ORR_ALLOW_STALE_LIB"):
                import warnings
                warnings.warn("libopenroborl_hip.so does NOT match the sources (built from %s, sources are %s) and could not be "
                              "rebuilt (%r); using it because ORR_ALLOW_STALE_LIB is set" % (library_hash(), source_hash(), e))
            else:
                raise RuntimeError("libopenroborl_hip.so is missing or stale and could not be built: %r "
                                   "(set ORR_ALLOW_STALE_LIB=1 to use a stale library anyway)" % (e,))
    # PyTorch-ROCm brings its own copy of the HIP runtime.  If this library is mapped FIRST it pulls in the system's libamdhip64, torch
    # then loads its own, and the process holds two runtimes - the second one to initialise finds "no ROCm-capable device" (seen with
    # `python __graft_entry__.py smoke`, where build() loaded the library before anything had imported torch).  Importing torch first
    # makes the dynamic loader resolve this library's HIP symbols against the runtime torch has already mapped.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    vp = C.c_void_p
    L.orr_last_error.restype = C.c_char_p
    L.orr_abi_version.restype = C.c_int32
    L.orr_source_hash.restype = C.c_char_p
    if path == LIB_PATH and not os.environ.get("ORR_ALLOW_STALE_LIB"):
        # what this process actually mapped (not what is on disk): must be the build of the sources on disk
        got, want = L.orr_source_hash().decode(), source_hash()
        if got != want:
            raise RuntimeError("loaded libopenroborl_hip.so was built from %s, the sources are %s (a stale mapping of a replaced "
                               "file?); set ORR_ALLOW_STALE_LIB=1 to run it anyway" % (got, want))
    L.orr_create.restype = C.c_int32
    L.orr_create.argtypes = [C.POINTER(_abi.OrrConfig), C.POINTER(vp)]
    L.orr_destroy.argtypes = [vp]
    L.orr_set_seed.restype = C.c_int32
    L.orr_set_seed.argtypes = [vp, C.c_uint64]
    L.orr_set_model.restype = C.c_int32
    L.orr_set_model.argtypes = [vp, C.c_int32, C.POINTER(_abi.OrrModel)]
    L.orr_set_motion.restype = C.c_int32
    L.orr_set_motion.argtypes = [vp, C.c_int32, vp, vp, C.c_int32, C.c_double, C.c_int32, C.POINTER(C.c_float)]
    L.orr_bind.restype = C.c_int32
    L.orr_bind.argtypes = [vp, vp, vp, vp, C.c_int32]
    L.orr_reset.restype = C.c_int32
    L.orr_reset.argtypes = [vp, vp, vp, vp]
    L.orr_step.restype = C.c_int32
    L.orr_step.argtypes = [vp, vp, vp, vp, vp, vp]
    L.orr_debug_physics.restype = C.c_int32
    L.orr_debug_physics.argtypes = [vp, vp, vp, C.c_int32, vp]
    L.orr_debug_replay_step.restype = C.c_int32
    L.orr_debug_replay_step.argtypes = [vp] * 10
    L.orr_debug_replay_reset.restype = C.c_int32
    L.orr_debug_replay_reset.argtypes = [vp, vp, vp, vp]
    L.orr_episode_stats.restype = C.c_int32
    L.orr_episode_stats.argtypes = [vp, C.c_double, C.c_int32, vp, vp]
    L.orr_time_steps.restype = C.c_int32
    L.orr_time_steps.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int32, C.POINTER(C.c_float)]
    L.orr_stress_actions.restype = C.c_int32
    L.orr_stress_actions.argtypes = [vp, vp, vp, vp, vp]
    L.orr_sizeof_config.restype = C.c_int32
    L.orr_sizeof_model.restype = C.c_int32
    # include/openroborl_policy.h
    L.orr_policy_packed_size.restype = C.c_int64
    L.orr_policy_packed_size.argtypes = [C.c_int32, C.c_int32]
    L.orr_policy_pack.restype = C.c_int32
    L.orr_policy_pack.argtypes = [vp, C.c_int32, C.c_int32, vp, vp]
    L.orr_gae.restype = C.c_int32
    L.orr_gae.argtypes = [vp, vp, vp, vp, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_int32, C.c_float, vp, vp, vp]
    L.orr_gae_flags.restype = C.c_int32
    L.orr_gae_flags.argtypes = [vp, vp, vp, vp, vp, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_int32, C.c_float, vp, vp, vp]
    L.orr_policy_forward.restype = C.c_int32
    L.orr_policy_forward.argtypes = [C.POINTER(_abi.OrrPolicyNet), vp, C.c_int32, vp, C.c_float, C.c_float, vp, vp, vp, vp, vp]
    # include/openroborl_learner.h
    L.orr_learner_workspace_floats.restype = C.c_int64
    L.orr_learner_workspace_floats.argtypes = [C.c_int32, C.c_int32]
    L.orr_ppo_head.restype = C.c_int32
    L.orr_ppo_head.argtypes = [vp, vp, vp, C.c_int32, C.c_float, C.c_float, C.c_float, vp, vp, vp, vp, vp, vp, vp]
    L.orr_relu_backward.restype = C.c_int32
    L.orr_relu_backward.argtypes = [vp, vp, C.c_int32, C.c_int32, vp, vp, vp]
    L.orr_head_backward.restype = C.c_int32
    L.orr_head_backward.argtypes = [vp, C.c_int32, vp, vp, C.c_int32, C.c_int32, vp, vp, vp, vp, vp]
    L.orr_learner_partial_rows.restype = C.c_int32
    L.orr_learner_partial_rows.argtypes = [C.c_int32]
    L.orr_colsum_finish.restype = C.c_int32
    L.orr_colsum_finish.argtypes = [C.POINTER(_abi.OrrColsumJob), C.c_int32, vp]
    L.orr_adam_step.restype = C.c_int32
    L.orr_adam_step.argtypes = [vp, vp, vp, vp, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32, vp, vp]
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
