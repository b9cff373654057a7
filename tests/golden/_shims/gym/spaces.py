"""gym.spaces.Box / Dict as plain containers (only .low / .high / .spaces are read by the reference)."""
import numpy as np


class Box(object):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.low = np.asarray(low, dtype=dtype)
        self.high = np.asarray(high, dtype=dtype)
        self.shape = self.low.shape
        self.dtype = np.dtype(dtype)


class Dict(object):
    def __init__(self, spaces):
        self.spaces = dict(spaces)

    def __getitem__(self, k):
        return self.spaces[k]
