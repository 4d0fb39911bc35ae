"""`gym.spaces` when gym is installed; otherwise a minimal Box/Space with the attributes the reference
algorithms read (`.shape`, `.high`, `.low`; ppo.py:34-39,57-61, ddpg.py:48)."""
import numpy as np

try:  # pragma: no cover - gym is absent in the build container
    from gym.spaces import Box, Space  # type: ignore
except Exception:
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

        def __repr__(self):
            return "Box(%s)" % (self.shape,)
