"""PCIe-inclusive rate of the reference's list-of-numpy protocol (LegacyListEnv) at the reference's batch sizes (development aid, GPU box).
Not the headline metric: inputs and outputs cross the host boundary every step."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openroborl_amd.env import LegacyListEnv, VecQuadrupedEnv  # noqa: E402

for n in (1, 2, 16, 256, 1024, 4096):
    env = VecQuadrupedEnv(num_robot=n, seed=1, robot="laikago", motion_file="laikago_pace", mode="test", auto_reset=False)
    leg = LegacyListEnv(env, mutate_actions=True)
    leg.reset()
    k = 200 if n <= 256 else 40
    for _ in range(5):
        leg.step([np.zeros(12, dtype=np.float32) for _ in range(n)])
    t0 = time.perf_counter()
    for _ in range(k):
        leg.step([np.zeros(12, dtype=np.float32) for _ in range(n)])
    dt = time.perf_counter() - t0
    print("LEGACY_HOST_RATE robots=%d env_steps_per_s=%.0f ms_per_step=%.3f" % (n, n * k / dt, 1e3 * dt / k), flush=True)
    env.close()
