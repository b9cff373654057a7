"""CPU-only checks of the host side: config, spaces pinned to the shipped policies' pickled bounds,
C-ABI library exports, state layout, error behaviour.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from openroborl_amd import _abi, _lib, config, env as envmod, motion, robots, state as statemod
from tests import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_observation_and_action_space_match_shipped_policies():
    """the 160-d bounds pickled inside policies/*.zip were produced by the real reference run."""
    g = np.load(os.path.join(ol.GOLDEN, "spaces.npz"))
    for pol, clip in (("laikago_pace", "laikago_pace"), ("minicheetah_trot", "minicheetah_trot")):
        sp = envmod.observation_space([motion.MotionClip(clip)])
        np.testing.assert_allclose(sp.low, g[pol + "/observation_space/low"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(sp.high, g[pol + "/observation_space/high"], rtol=1e-6, atol=1e-6)
        a = envmod.action_space()
        np.testing.assert_allclose(a.low, g[pol + "/action_space/low"], rtol=1e-6)
        np.testing.assert_allclose(a.high, g[pol + "/action_space/high"], rtol=1e-6)
        assert sp.shape == (160,) and a.shape == (12,)


def test_sensor_history_layout_matches_reference():
    """A5 / D3: flattened order IMU | LastAction | MotorAngle, 3-deep, newest first (golden from the reference's
    own sensor classes), and the same bounds."""
    g = np.load(os.path.join(ol.GOLDEN, "sensors.npz"))
    assert [str(n) for n in g["names"]] == ["HistoricSensorWrapper(IMU)", "HistoricSensorWrapper(LastAction)",
                                            "HistoricSensorWrapper(MotorAngle)"]
    feeds, obs = g["feeds"], g["obs"]          # feed = [angles 12 | rpy 3 | drpy 3 | last_action 12]
    hist = {"imu": None, "act": None, "ang": None}

    def reading(f):
        return np.array([f[12], f[13], f[15], f[16]]), f[18:30], f[0:12]
    imu, act, ang = reading(feeds[0])
    H = [np.tile(imu, 3), np.tile(act, 3), np.tile(ang, 3)]         # on_reset: 3 copies
    np.testing.assert_allclose(np.concatenate(H), obs[0])
    for k in range(1, feeds.shape[0]):
        new = reading(feeds[k])
        for i, w in enumerate((4, 12, 12)):
            H[i] = np.concatenate([new[i], H[i][:2 * w]])             # newest first
        np.testing.assert_allclose(np.concatenate(H), obs[k])
    lo, hi = envmod.proprio_bounds()
    np.testing.assert_allclose(lo, g["low"])
    np.testing.assert_allclose(hi, g["high"])


def test_yaml_schema_and_config():
    for task in config.TASKS:
        p = config.load_training_params(task)
        for key in ("robot", "seed", "num_robot", "mode", "output_dir", "num_test_episodes", "total_timesteps",
                    "int_save_freq", "timestep_per_actorbach", "optim_batchsize", "enable_env_randomizer",
                    "motion_file", "model_file"):
            assert key in p, key
    sim = config.load_sim_params()
    assert sim["sim_time_step_s"] == 0.001 and sim["num_sim_iter_step"] == 300
    c = config.make_config(8, sim_params=sim, mode="train")
    assert c.solver_iters == 9 and c.action_repeat == 33 and abs(c.gravity_z + 10.0) < 1e-9
    assert c.flags & _abi.FLAG_RANDOMIZER and c.ep_len_start == 20 and c.ep_len_end == 600
    assert not (config.make_config(8, mode="test").flags & _abi.FLAG_RANDOMIZER)
    with pytest.raises(ValueError):
        config.load_training_params("hybrid_gait_minicheetah")


def test_library_loads_and_exports_every_declared_symbol():
    L = _lib.load()
    hdr = "".join(open(os.path.join(ROOT, "include", h)).read() for h in ("openroborl_hip.h", "openroborl_policy.h", "openroborl_learner.h"))
    declared = set(re.findall(r"\b(orr_[a-z_]+)\s*\(", hdr))
    declared -= {"orr_handle"}
    assert declared, "no declarations found"
    for name in sorted(declared):
        assert hasattr(L, name), "missing export %s" % name
    assert set(_lib.EXPORTS) <= declared
    assert L.orr_sizeof_config() == C.sizeof(_abi.OrrConfig)
    assert L.orr_sizeof_model() == C.sizeof(_abi.OrrModel)


def test_layout_is_shared_by_library_and_oracle():
    a = statemod.Layout(_lib.load(), "orr")
    b = ol.layout()
    assert a.fields == b.fields and a.stride == b.stride == _abi.STATE_STRIDE
    end = max(off + size for off, size, _ in a.fields.values())
    assert end <= a.stride
    assert a.fields["RING"][1] == _abi.RING_DEPTH * _abi.RING_ENTRY


def test_bad_arguments_are_reported_not_thrown():
    L = _lib.load()
    cfg = config.make_config(4)
    cfg.abi_version = 99
    h = C.c_void_p()
    assert L.orr_create(C.byref(cfg), C.byref(h)) < 0
    assert b"ABI" in L.orr_last_error()
    cfg = config.make_config(4)
    cfg.num_robots = 0
    assert L.orr_create(C.byref(cfg), C.byref(h)) < 0
    # the quaternion step of the kernel is a series in the rotation of one sub-step: a velocity clamp that would let the base turn
    # more than 0.4 rad per sub-step is refused
    cfg = config.make_config(4)
    cfg.max_coord_velocity = 1000.0
    assert L.orr_create(C.byref(cfg), C.byref(h)) < 0 and b"max_coord_velocity" in L.orr_last_error()
    # no GPU in the build container: creation must fail loudly instead of falling back to the CPU
    import torch
    if not torch.cuda.is_available():
        cfg = config.make_config(4)
        rc = L.orr_create(C.byref(cfg), C.byref(h))
        assert rc < 0 and b"no HIP device" in L.orr_last_error()
        with pytest.raises(RuntimeError):
            envmod.VecQuadrupedEnv(task_name="imitation_learning_laikago", num_robot=2)


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "openroborl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "liborr_oracle" not in txt and "orc_" not in txt and "oracle_lib" not in txt, f


def test_motion_clip_validator():
    from openroborl_amd import motion
    rep, problems = motion.validate("laikago_pace")
    assert rep["frames"] > 10 and rep["loop"] == "Wrap" and not problems
    import json, tempfile
    js = json.load(open(motion.resolve_path("laikago_pace")))
    js["Frames"][3][7] += 5.0                       # a wrapped angle in one frame
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
        json.dump(js, f)
    rep, problems = motion.validate(f.name)
    assert any("joint rate" in p for p in problems)
    os.unlink(f.name)


def test_headers_are_plain_c():
    """The drop-in boundary is a C ABI: the headers must compile as C99 on their own."""
    import subprocess
    for h in ("openroborl_hip.h", "openroborl_policy.h", "openroborl_learner.h"):
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-fsyntax-only", "-x", "c", os.path.join(ROOT, "include", h)])


@pytest.mark.parametrize("defs", [
    ["-DORR_GENERIC_PGS"], ["-DORR_PHASE_TIMERS"], ["-DORR_COUNT_DUAL_CONTACT"], ["-DORR_WPB=2"], ["-DORR_WAVE_TIMELINE"], ["-DORR_WAVES_PER_EU=2"]])
def test_the_kernel_tuning_knobs_still_compile(defs):
    """What is left of the step kernel's preprocessor switches after round 6's clean-up (the A/B alternatives that lost are gone from the
    source; their measurements stay in HISTORY.md / profiles/r04_ab*): the development aids - the readable twin of the hand-scheduled PGS
    block, the phase timers, the dual-contact counter, the wave timeline - and the two launch-shape experiments the tools still drive
    (tools/wave_pairing.py).  Each goes through the device compiler's front end here (syntax + templates + static_asserts; well under a
    second each), in all three translation units."""
    import subprocess
    base = [f for f in _lib.HIPCC_FLAGS if f not in ("-shared", "-fPIC")] + ["--cuda-device-only", "-fsyntax-only", "-Wno-unused-command-line-argument"]
    csrc = os.path.dirname(_lib.SRC)
    for src in (_lib.SRC, os.path.join(csrc, "orr_kernels_w2.hip"), os.path.join(csrc, "orr_kernels_anchor.hip")):
        r = subprocess.run([_lib.HIPCC] + base + defs + [src], capture_output=True, text=True)
        assert r.returncode == 0, "%s %s:\n%s" % (os.path.basename(src), " ".join(defs), r.stderr[-1500:])


def test_policy_abi_argument_checks_without_gpu():
    """Host-side argument validation of include/openroborl_policy.h (no launch happens on these paths)."""
    L = _lib.load()
    assert L.orr_policy_packed_size(160, 512) == 32 * 10 * 256
    assert L.orr_policy_packed_size(256, 12) == 1 * 16 * 256          # 12 columns padded to one 16-column tile
    assert L.orr_policy_packed_size(100, 12) == -1                    # K must be a multiple of 16
    assert L.orr_policy_pack(None, 160, 512, None, None) == -1
    assert b"orr_policy_pack" in L.orr_last_error()
    net = _abi.OrrPolicyNet()
    assert L.orr_policy_forward(C.byref(net), None, 4, None, 0.125, 6.28, None, None, None, None, None) == -1
    assert b"orr_policy_forward" in L.orr_last_error()
    assert L.orr_gae(None, None, None, None, 4, 4, 0.95, 0.95, 1, 0.0, None, None, None) == -1
    assert b"orr_gae" in L.orr_last_error()


def test_learner_abi_argument_checks_without_gpu():
    """Host-side argument validation of include/openroborl_learner.h (no launch happens on these paths)."""
    L = _lib.load()
    assert L.orr_learner_workspace_floats(16384, 256) == 512 * 256 * 13       # 32 rows per workgroup: [rows][c] bias + [rows][c][12] weight partials
    assert L.orr_learner_workspace_floats(64, 4) == 2 * 4 * 13
    assert L.orr_learner_partial_rows(16384) == 512 and L.orr_learner_partial_rows(33) == 2 and L.orr_learner_partial_rows(0) == -1
    assert L.orr_learner_workspace_floats(0, 512) == -1
    assert L.orr_ppo_head(None, None, None, 16, 0.125, 0.2, 1.0, None, None, None, None, None, None, None) == -1
    assert b"orr_ppo_head" in L.orr_last_error()
    assert L.orr_relu_backward(None, None, 16, 512, None, None, None) == -1
    assert L.orr_relu_backward(16, 16, 16, 24, 16, 16, None) == -1            # 24 / 4 = 6 columns of four do not divide a workgroup
    assert b"multiple of 4" in L.orr_last_error()
    assert L.orr_head_backward(16, 3, 16, 16, 16, 256, 16, 16, 16, 16, None) == -1
    assert b"fan-out" in L.orr_last_error()
    assert L.orr_head_backward(16, 12, 16, 16, 16, 256, 16, 16, None, 16, None) == -1
    assert b"both" in L.orr_last_error()
    assert L.orr_colsum_finish(None, 1, None) == -1
    jobs = (_abi.OrrColsumJob * 13)()
    assert L.orr_colsum_finish(jobs, 13, None) == -1
    assert L.orr_colsum_finish(jobs, 1, None) == -1                           # a job with null pointers
    assert b"bad job" in L.orr_last_error()
    assert L.orr_adam_step(16, 16, 16, 16, 8, 1e-4, 0.9, 0.999, 1e-5, 1.0, 2, 16, None) == -1
    assert b"unknown flag" in L.orr_last_error()
    assert L.orr_adam_step(16, 16, 16, 20, 8, 1e-4, 0.9, 0.999, 1e-5, 1.0, 0, 16, None) == -1
    assert b"aligned" in L.orr_last_error()


def test_stale_library_is_rebuilt_and_the_new_build_is_what_gets_loaded(tmp_path):
    """Loader regression: the stale-library check must not dlopen the stale file (ctypes never dlcloses and glibc
    returns the old mapping for the same path).  A copy of the package with an edited dependency and the OLD library in place:
    load() in one fresh process must rebuild, and the library it has mapped must carry the NEW hash."""
    import shutil
    import subprocess
    import sys
    assert _lib.library_hash() == _lib.source_hash(), "in-tree library is stale"
    pkg = tmp_path / "openroborl_amd"
    shutil.copytree(os.path.join(ROOT, "openroborl_amd"), pkg, ignore=shutil.ignore_patterns("__pycache__", "data", "*.o", "lib_ab_old.so"))
    shutil.copytree(os.path.join(ROOT, "include"), tmp_path / "include")
    with open(pkg / "csrc" / "orr_task.h", "a") as f:
        f.write("\n// edited by the loader test\n")
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from openroborl_amd import _lib\n"
            "old = _lib.library_hash(); want = _lib.source_hash()\n"
            "assert old is not None and old != want and _lib.needs_build()\n"
            "L = _lib.load()\n"
            "got = L.orr_source_hash().decode()\n"
            "assert got == want == _lib.library_hash(), (got, want, old)\n"
            "print('LOADED', got, 'OLD', old)\n" % str(tmp_path))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                         env={k: v for k, v in os.environ.items() if not k.startswith("ORR_")})
    assert out.returncode == 0, out.stdout + out.stderr
    assert "LOADED" in out.stdout and _lib.source_hash() in out.stdout.split("OLD")[1]


def test_tuning_defines_are_part_of_the_source_hash(monkeypatch):
    base = _lib.source_hash()
    monkeypatch.setenv("ORR_EXTRA_DEFS", "ORR_SOMETHING=1")
    assert _lib.source_hash() != base and _lib.needs_build()
    monkeypatch.delenv("ORR_EXTRA_DEFS")
    monkeypatch.setenv("ORR_WAVES_PER_EU", "2")
    assert _lib.source_hash() != base
    monkeypatch.delenv("ORR_WAVES_PER_EU")
    assert _lib.source_hash() == base and _lib.source_hash(("-DX",)) != base


def test_null_handles_are_reported_not_dereferenced():
    """Without a device no handle exists here: the NULL-handle paths of bind / step must report instead of crashing (the alignment
    checks behind them need a real handle: tests/test_gpu_parity.py::test_misaligned_buffers_are_refused)."""
    L = _lib.load()
    assert L.orr_bind(None, C.c_void_p(16), C.c_void_p(16), None, 0) < 0 and b"orr_bind" in L.orr_last_error()
    assert L.orr_step(None, None, None, None, None, None) < 0


def test_graft_entry_build_checks_the_current_abi_version():
    """__graft_entry__.build() is the driver's "does it build" check: it must assert the ABI version the header declares (round 5 bumped the
    header to v5 while build() still asserted 4 - caught before the driver ran it)."""
    import re
    hdr = open(os.path.join(ROOT, "include", "openroborl_hip.h")).read()
    ver = int(re.search(r"#define ORR_ABI_VERSION (\d+)", hdr).group(1))
    from openroborl_amd import _abi
    assert _abi.ABI_VERSION == ver
    src = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert "_abi.ABI_VERSION == %d" % ver in src
