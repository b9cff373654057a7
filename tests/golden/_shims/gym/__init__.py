"""Import shim for the golden-fixture generators (tests/golden/make_golden_task.py): the handful of `gym` names the
reference's env modules touch at import / construction time (quadruped_gym_env.py:19-21, minitaur.py:27, wrapper_env.py:22).
gym is not installed in this image.  Test infrastructure only; never imported by the product package."""
from . import spaces  # noqa: F401


class Env(object):
    pass
