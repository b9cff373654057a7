"""Import shim: module-level constants of pybullet read by quadruped_gym_env.py:173-184.  The client object itself is
tests/golden/fake_bullet.py (a scripted stand-in, NOT a physics engine)."""
GUI = 1
DIRECT = 2
