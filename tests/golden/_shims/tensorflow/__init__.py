"""Import shim: the randomiser only calls tf.logging.info (controllable_env_randomizer_from_config.py:50)."""


class _Logging(object):
    @staticmethod
    def info(*a, **k):
        pass


logging = _Logging()
