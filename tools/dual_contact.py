"""How often do the toe sphere AND the shank sphere of a lower leg touch the ground in the same sub-step? (GPU box)

Lower legs are feet (minitaur.py:842-844, imitation_task.py:536-546): Bullet makes a contact point for EVERY touching shape, this engine
takes the lower of the leg's two spheres (one contact point per leg and sub-step).  The -DORR_COUNT_DUAL_CONTACT
build counts, per leg and sub-step, which spheres are within the contact margin (0.02 m: Bullet's contact breaking threshold, where
a contact row exists) and which penetrate.  Workloads: the bench's stress actions (train semantics, 4096 Laikago robots), the soak's
random actions (sigma 0.3 rad), the mixed Laikago + mini-cheetah batch, a train.py run from scratch, and the shipped laikago_pace /
minicheetah_trot policies in test mode.

usage (GPU box):  python tools/dual_contact.py [steps=2000] [train_iters=300]
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "openroborl_amd", "libopenroborl_dual_contact.so")
from openroborl_amd import _lib as _build  # noqa: E402  (build only; the library is loaded below)
_build.build(out_path=LIB, extra_flags=["-DORR_COUNT_DUAL_CONTACT"])
os.environ["ORR_LIB_PATH"] = LIB
os.environ["ORR_STEP_WAVES_PER_EU"] = "1"      # the counters live in the one-wave kernel

import torch  # noqa: E402
from openroborl_amd import _lib, policy as polmod  # noqa: E402
from openroborl_amd.env import VecQuadrupedEnv  # noqa: E402

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
TRAIN_ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 300
L = _lib.load()
L.orr_debug_dual_contact.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * 8)()
NAMES = ("leg_substeps", "with_contact_row", "both_within_margin", "both_penetrating", "shank_only_within_margin", "row_at_shank")
rows = []


def read(label):
    L.orr_debug_dual_contact(buf, 1)
    v = dict(zip(NAMES, [int(x) for x in buf[:6]]))
    tot = max(v["leg_substeps"], 1)
    v.update(label=label, frac_both_within_margin=v["both_within_margin"] / tot, frac_both_penetrating=v["both_penetrating"] / tot,
             frac_row_at_shank=v["row_at_shank"] / tot, frac_with_contact_row=v["with_contact_row"] / tot)
    rows.append(v)
    print("%-46s leg-sub-steps %.3e | contact row %.3f | both within margin %.3e | both penetrating %.3e | row at the shank sphere %.3e"
          % (label, v["leg_substeps"], v["frac_with_contact_row"], v["frac_both_within_margin"], v["frac_both_penetrating"], v["frac_row_at_shank"]), flush=True)


def stress(task, n, label, **kw):
    env = VecQuadrupedEnv(task_name=task, num_robot=n, mode="train", auto_reset=True, seed=0, **kw)
    obs = env.reset()
    gen = torch.Generator(device=env.device).manual_seed(1)
    noise = torch.empty(n, 12, device=env.device)
    act = torch.empty(n, 12, device=env.device)
    L.orr_debug_dual_contact(buf, 1)
    for _ in range(STEPS):
        noise.normal_(generator=gen).mul_(0.125)
        obs = env.step(env.stress_actions(env.obs, noise, act))[0]
    read(label)
    env.close()


def soak(n=4096):
    env = VecQuadrupedEnv(task_name="imitation_learning_laikago", num_robot=n, mode="train", auto_reset=True, seed=7)
    env.reset()
    g = torch.Generator(device=env.device).manual_seed(0)
    L.orr_debug_dual_contact(buf, 1)
    for _ in range(STEPS):
        env.step(torch.randn(n, 12, device=env.device, generator=g) * 0.3)
    read("soak: random actions sigma 0.3, %d steps" % STEPS)
    env.close()


def shipped(pol, clip, robot, n=1024):
    env = VecQuadrupedEnv(num_robot=n, seed=1, robot=robot, motion_file=clip, mode="test", enable_randomizer=False, auto_reset=True)
    model = polmod.MLPPolicy.from_file(os.path.join(ROOT, "tests", "golden", "policy_%s.npz" % pol), env.device)
    obs = env.reset()
    L.orr_debug_dual_contact(buf, 1)
    for _ in range(600):
        obs = env.step(model.act(obs, deterministic=True)[0].contiguous())[0]
    read("shipped %s policy, test mode, 600 steps" % pol)
    env.close()


stress("imitation_learning_laikago", 4096, "bench stress actions, Laikago, %d steps" % STEPS)
stress("imitation_learning_minicheetah", 4096, "bench stress actions, mini-cheetah, %d steps" % STEPS)
soak()
shipped("laikago_pace", "laikago_pace", "laikago")
shipped("minicheetah_trot", "minicheetah_trot", "mini_cheetah")
# a training run from scratch in this process (train.py's main), counted over all its env steps
import train  # noqa: E402
sys.argv = ["train.py", "--num-robot", "4096", "--iters", str(TRAIN_ITERS)]
L.orr_debug_dual_contact(buf, 1)
train.main()
read("train.py from scratch, Laikago pace, %d iterations x 32 steps" % TRAIN_ITERS)
out = os.path.join(ROOT, "gpurun_out", "dual_contact.json")
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump(rows, open(out, "w"), indent=1)
print("written", out)
