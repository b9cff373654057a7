"""gym.utils.seeding.np_random (quadruped_gym_env.py:59-61; the returned generator is never used by the reference)."""
import numpy as np


def np_random(seed=None):
    return np.random.RandomState(seed), seed
