import numpy as np


class Space:
    pass


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            low = np.asarray(low, dtype=dtype)
            high = np.asarray(high, dtype=dtype)
            shape = low.shape
        else:
            low = np.full(shape, low, dtype=dtype)
            high = np.full(shape, high, dtype=dtype)
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype
