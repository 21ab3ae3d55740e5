"""Gaussian record and Gaussian fields -- host mirror of the reference's src/rtgs/gaussian.py.

The reference's ``Gaussian`` is a 236-byte AoS Taichi struct (gaussian.py:26-55).  Here a
``GaussianField`` is structure-of-arrays on the HIP device (one tensor per member), which is
what the render kernels stream; ``field[i]`` still returns a ``Gaussian`` record with the
reference's member names.  The per-Gaussian methods (cov / hit / eval / eval_sh) are device
functions in the reference and run HIP kernels here (include/rtgs_hip.h); there is no CPU path.
"""

from __future__ import annotations

import math
from typing import NamedTuple

import numpy as np
import torch

from . import _native as nat
from .ray import Ray
from .utils.types import vec2, vec3, vec4

BOUNDING_THRESHOLD = 3  # gaussian.py:13: squared Mahalanobis radius of the hit ellipsoid


class Bound(NamedTuple):
    """bounding_box.py:5-13: (p_min, p_max) of an axis-aligned box."""
    p_min: np.ndarray
    p_max: np.ndarray

    def get_centroid(self):      # bounding_box.py:21-22
        return vec3(0.5 * (self.p_min + self.p_max))

    def size(self):              # bounding_box.py:24-26
        return vec3(self.p_max - self.p_min)

SH_NAMES = ("sh_10", "sh_11", "sh_12", "sh_20", "sh_21", "sh_22", "sh_23", "sh_24",
            "sh_30", "sh_31", "sh_32", "sh_33", "sh_34", "sh_35", "sh_36")  # gaussian.py:41-55


def _t(x, shape, dev):
    return torch.as_tensor(np.asarray(x, dtype=np.float32).reshape(shape)).to(dev).contiguous()


class Gaussian:
    """gaussian.py:26-55.  ``Gaussian()`` is zero-initialised, like a Taichi dataclass."""

    __slots__ = ("position", "rotation", "scale", "color", "opacity") + SH_NAMES

    def __init__(self, position=(0, 0, 0), rotation=(0, 0, 0, 0), scale=(0, 0, 0),
                 color=(0, 0, 0), opacity=0.0, **sh):
        self.position = vec3(position)
        self.rotation = vec4(rotation)
        self.scale = vec3(scale)
        self.color = vec3(color)
        self.opacity = float(np.float32(opacity))
        for name in SH_NAMES:
            setattr(self, name, vec3(sh.pop(name, (0, 0, 0))))
        if sh:
            raise TypeError(f"unknown Gaussian members: {sorted(sh)}")

    def init(self, position=(0, 0, 0), rotation=(0, 0, 0, 1), scale=(1, 1, 1), color=(1, 0, 1),
             opacity=1):
        """gaussian.py:59-84: unit Gaussian at the origin, magenta, opaque.  SH untouched."""
        self.position = vec3(position)
        self.rotation = vec4(rotation)
        self.scale = vec3(scale)
        self.color = vec3(color)
        self.opacity = float(np.float32(opacity))

    # ---- helpers -------------------------------------------------------------------
    def sh_array(self) -> np.ndarray:
        return np.stack([getattr(self, n) for n in SH_NAMES]).astype(np.float32)  # (15, 3)

    def _cov_pair(self):
        dev = nat.device()
        rot = _t(self.rotation, (1, 4), dev)
        sc = _t(self.scale, (1, 3), dev)
        cov = torch.empty((1, 6), dtype=torch.float32, device=dev)
        inv = torch.empty((1, 6), dtype=torch.float32, device=dev)
        nat.check(nat.lib().rtgs_covariance(1, nat.ptr(rot), nat.ptr(sc), nat.ptr(cov),
                                            nat.ptr(inv), nat.stream()), "rtgs_covariance")
        return cov, inv

    # ---- device functions of the reference ----------------------------------------------
    def cov(self) -> np.ndarray:
        """R S S^T R^T (gaussian.py:86-102) as a 3x3 array."""
        c = self._cov_pair()[0][0].cpu().numpy()
        return np.array([[c[0], c[1], c[2]], [c[1], c[3], c[4]], [c[2], c[4], c[5]]], dtype=np.float32)

    def hit(self, ray: Ray) -> vec2:
        """Ray/ellipsoid intersections (t1, t2), both inf on a miss (gaussian.py:203-230)."""
        dev = nat.device()
        rays = torch.from_numpy(ray.as_array()[None, :]).to(dev)
        pos = _t(self.position, (1, 3), dev)
        rot, sc = _t(self.rotation, (1, 4), dev), _t(self.scale, (1, 3), dev)
        out = torch.empty((1, 2), dtype=torch.float32, device=dev)
        nat.check(nat.lib().rtgs_gaussian_hit(1, nat.ptr(rays), nat.ptr(pos), nat.ptr(rot), nat.ptr(sc),
                                              nat.ptr(out), nat.stream()), "rtgs_gaussian_hit")
        return vec2(out[0].cpu().numpy())

    def _eval(self, pos, direction, color, opacity, with_sh=True) -> np.ndarray:
        dev = nat.device()
        rot, sc = _t(self.rotation, (1, 4), dev), _t(self.scale, (1, 3), dev)
        p = _t(pos, (1, 3), dev)
        d = _t(direction, (1, 3), dev)
        mu = _t(self.position, (1, 3), dev)
        colop = _t(list(color) + [opacity], (1, 4), dev)
        sh = _t(self.sh_array(), (1, 45), dev) if with_sh else None
        out = torch.empty((1, 4), dtype=torch.float32, device=dev)
        nat.check(nat.lib().rtgs_gaussian_eval(1, nat.ptr(p), nat.ptr(d), nat.ptr(mu), nat.ptr(rot),
                                               nat.ptr(sc), nat.ptr(colop), nat.ptr(sh), nat.ptr(out),
                                               nat.stream()), "rtgs_gaussian_eval")
        return out[0].cpu().numpy()

    def eval(self, pos, dir) -> vec4:  # noqa: A002 - the reference's parameter name
        """(rgb, alpha) at pos seen along dir (gaussian.py:183-201)."""
        return vec4(self._eval(pos, dir, self.color, self.opacity))

    def eval_sh(self, dir) -> vec3:  # noqa: A002
        """Degree 1..3 SH radiance for a NORMALISED dir (gaussian.py:140-181)."""
        return vec3(self._eval(self.position, dir, (0, 0, 0), 1.0)[:3])

    def bounding_box(self) -> "Bound":
        """Axis-aligned box of the six end points position +- R S (3 e_k) of the ellipsoid's axes
        (gaussian.py:104-138; BOUNDING_THRESHOLD = 3 scale units, conservative-but-loose by the reference's own
        TODO at :112).  Only the reference's BVH build consumes it -- this path culls with per-frame screen bounds
        and exact world boxes of its own (``Scene.accel``) -- but it is on the surface a caller sees.  The rotation
        matrix comes from the device (``quat.as_rotation_mat3``: rtgs_quat_op), like every other method here."""
        from .utils import quaternion as quat
        r = quat.as_rotation_mat3(np.asarray(self.rotation, dtype=np.float32)).cpu().numpy().astype(np.float32)
        arms = r * (np.float32(BOUNDING_THRESHOLD) * np.asarray(self.scale, dtype=np.float32))[None, :]   # column k = R S (3 e_k)
        p = np.asarray(self.position, dtype=np.float32)
        pts = np.concatenate([p[None, :] + arms.T, p[None, :] - arms.T], axis=0)
        return Bound(vec3(pts.min(axis=0)), vec3(pts.max(axis=0)))

    @staticmethod
    def field(shape) -> "GaussianField":
        return GaussianField(shape)

    def __repr__(self):
        return (f"Gaussian(position={self.position.tolist()}, rotation={self.rotation.tolist()}, "
                f"scale={self.scale.tolist()}, color={self.color.tolist()}, opacity={self.opacity})")


class GaussianField:
    """Structure-of-arrays Gaussians on the HIP device; zero-initialised like ``Gaussian.field``."""

    def __init__(self, shape, device=None):
        self.shape = () if shape == () else ((shape,) if isinstance(shape, int) else tuple(int(s) for s in shape))
        n = math.prod(self.shape) if self.shape else 1
        dev = device or nat.device()
        z = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)  # noqa: E731
        self.n = n
        self.position, self.rotation, self.scale = z(n, 3), z(n, 4), z(n, 3)
        self.color, self.opacity, self.sh = z(n, 3), z(n), z(n, 15, 3)

    def init(self, position=(0, 0, 0), rotation=(0, 0, 0, 1), scale=(1, 1, 1), color=(1, 0, 1),
             opacity=1):
        """Run ``Gaussian.init`` (gaussian.py:59-84) on every element."""
        dev = self.position.device
        self.position[:] = _t(position, (3,), dev)
        self.rotation[:] = _t(rotation, (4,), dev)
        self.scale[:] = _t(scale, (3,), dev)
        self.color[:] = _t(color, (3,), dev)
        self.opacity[:] = float(opacity)

    def _flat(self, idx) -> int:
        if self.shape == ():
            return 0
        if isinstance(idx, tuple):
            return int(np.ravel_multi_index(idx, self.shape))
        return int(idx) % self.n if self.n else 0

    def __getitem__(self, idx) -> Gaussian:
        k = self._flat(idx)
        sh = self.sh[k].cpu().numpy()
        return Gaussian(self.position[k].cpu().numpy(), self.rotation[k].cpu().numpy(),
                        self.scale[k].cpu().numpy(), self.color[k].cpu().numpy(),
                        float(self.opacity[k]), **{n: sh[i] for i, n in enumerate(SH_NAMES)})

    def __setitem__(self, idx, g: Gaussian):
        k = self._flat(idx)
        dev = self.position.device
        self.position[k] = _t(g.position, (3,), dev)
        self.rotation[k] = _t(g.rotation, (4,), dev)
        self.scale[k] = _t(g.scale, (3,), dev)
        self.color[k] = _t(g.color, (3,), dev)
        self.opacity[k] = g.opacity
        self.sh[k] = _t(g.sh_array(), (15, 3), dev)

    def __len__(self):
        return self.n


def new_gaussian(position=(0, 0, 0), rotation=(0, 0, 0, 1), scale=(1, 1, 1), color=(1, 0, 1),
                 opacity=1) -> Gaussian:
    """Python-scope constructor (gaussian.py:233-247)."""
    return Gaussian(position, rotation, scale, color, opacity)
