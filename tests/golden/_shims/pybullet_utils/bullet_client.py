"""Import shim for quadruped_gym_env.py:24: BulletClient -> the scripted client of tests/golden/fake_bullet.py."""
from fake_bullet import FakeBulletClient as BulletClient  # noqa: F401
