"""Small vector constructors standing in for the reference's Taichi vector types
(src/rtgs/utils/types.py:1-4 vec2i/vec3i; ti.math.vec2/3/4 elsewhere).  They are plain NumPy
arrays with .x/.y/.z/.w accessors, enough for the host-side API."""

import numpy as np


class _Vec(np.ndarray):
    _n = 0
    _dtype = np.float32

    def __new__(cls, *args):
        if len(args) == 1 and np.ndim(args[0]) > 0:
            vals = np.asarray(args[0], dtype=cls._dtype).reshape(-1)
        elif len(args) == 1:
            vals = np.full(cls._n, args[0], dtype=cls._dtype)
        elif len(args) == 0:
            vals = np.zeros(cls._n, dtype=cls._dtype)
        else:
            vals = np.asarray(args, dtype=cls._dtype)
        if vals.shape != (cls._n,):
            raise ValueError(f"{cls.__name__} takes {cls._n} components")
        return vals.view(cls)

    x = property(lambda s: s[0].item())
    y = property(lambda s: s[1].item())
    z = property(lambda s: s[2].item())
    w = property(lambda s: s[3].item())

    def __eq__(self, other):  # value equality, as the reference's tests use it
        return bool(np.array_equal(np.asarray(self), np.asarray(other, dtype=self.dtype).reshape(-1)))

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None


class vec2(_Vec):
    _n = 2


class vec3(_Vec):
    _n = 3


class vec4(_Vec):
    _n = 4


class vec2i(_Vec):
    _n = 2
    _dtype = np.int32


class vec3i(_Vec):
    _n = 3
    _dtype = np.int32
