#!/usr/bin/env python3
"""Re-serialise the reference's motion clips (data, JSON) into openroborl_amd/data/motions/.

The clips are mocap-retargeted DATA files (19-float frames: root pos 3, root quat xyzw 4, 12 joint
angles; reference: OpenRoboRL/envs/quadruped_robot/task/motions/*.txt).  The env needs them at run
time and /root/reference does not exist on the GPU box, so they ship with the package.  The file
format is unchanged (same keys), so a user's own clip in the reference format loads as well.
"""
import json
import os
import sys

SRC = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/OpenRoboRL/envs/quadruped_robot/task/motions"
DST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "openroborl_amd", "data", "motions")
os.makedirs(DST, exist_ok=True)
for f in sorted(os.listdir(SRC)):
    if not f.endswith(".txt"):
        continue
    with open(os.path.join(SRC, f)) as fh:
        clip = json.load(fh)
    with open(os.path.join(DST, f), "w") as fh:
        json.dump(clip, fh, separators=(",", ":"))
    print(f, len(clip["Frames"]))
