"""Import shim for quadruped_gym_env.py:25,185 (pybullet_data.getDataPath)."""


def getDataPath():
    return "<pybullet_data is not installed>"
