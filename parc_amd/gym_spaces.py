"""Minimal stand-in for ``gym.spaces.Box`` (gym is not a dependency of the MI355X build).

The reference only reads ``.shape .dtype .low .high`` from the spaces the env returns and checks
``isinstance(space, gym.spaces.Box)`` (learning/base_agent.py:106,199,459); the agent in this package checks against
this class instead.  If the real gym is importable its Box is used, so foreign code keeps working.
"""
import numpy as np

try:  # pragma: no cover - gym is absent in the build image
    from gym.spaces import Box, Discrete  # type: ignore
except Exception:  # noqa: BLE001
    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):
            if shape is None:
                low = np.asarray(low, dtype=dtype)
                high = np.asarray(high, dtype=dtype)
                shape = low.shape
            else:
                shape = tuple(int(s) for s in shape)
                low = np.full(shape, low, dtype=dtype) if np.isscalar(low) else np.asarray(low, dtype=dtype)
                high = np.full(shape, high, dtype=dtype) if np.isscalar(high) else np.asarray(high, dtype=dtype)
            self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), np.dtype(dtype)

        def __repr__(self):
            return "Box({}, {})".format(self.shape, self.dtype)

    class Discrete:
        def __init__(self, n):
            self.n = int(n)
            self.shape = ()
            self.dtype = np.dtype(np.int64)
