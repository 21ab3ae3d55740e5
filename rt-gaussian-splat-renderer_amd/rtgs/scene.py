"""Gaussian-splat scene -- host mirror of the reference's src/rtgs/scene.py.

The reference's ``Scene`` owns an AoS Gaussian field plus a binned-SAH BVH that accelerates
``Scene.hit`` (scene.py:162-450).  Here the scene owns structure-of-arrays device buffers
(position, rotation, scale, colour+opacity, optional SH) that the HIP kernels stream; the
closest-hit *result* of ``Scene.hit`` (scene.py:433-437) is kept, the BVH is not built
(SURVEY.md 8(f) #3): per-frame screen-space culling replaces it on the render path.
"""

from __future__ import annotations

import logging
import os
import pathlib

import numpy as np
import torch

from . import _native as nat
from .gaussian import GaussianField
from .ray import Ray, RayField
from .utils.math import sigmoid
from .utils.types import vec2

logger = logging.getLogger(__name__)

HUGE_LAST = os.environ.get("RTGS_HUGE_LAST", "1") != "0"   # see morton_order

_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2",
              "int16": "i2", "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4",
              "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8",
              "float64": "f8"}


class SceneHit:
    """scene.py:24-33.  gaussian_idx == -1 means miss; depth is the reference's BVH leaf depth
    (debug only) and is always 0 here."""

    __slots__ = ("gaussian_idx", "intersections", "depth")

    def __init__(self, gaussian_idx=-1, intersections=(np.inf, np.inf), depth=0):
        self.gaussian_idx = int(gaussian_idx)
        self.intersections = vec2(intersections)
        self.depth = int(depth)

    def __repr__(self):
        return f"SceneHit(gaussian_idx={self.gaussian_idx}, intersections={self.intersections.tolist()})"


def read_ply(path) -> dict:
    """Vertex element of an ASCII/binary-little-endian PLY as {property: ndarray}."""
    data = pathlib.Path(path).read_bytes()
    end, marker = -1, b""
    for marker in (b"end_header\n", b"end_header\r\n"):
        end = data.find(marker)
        if end >= 0:
            break
    if not data.startswith(b"ply") or end < 0:
        raise ValueError(f"{path}: not a PLY file")
    lines = data[:end].decode("ascii").splitlines()
    fmt = next((ln.split()[1] for ln in lines if ln.startswith("format")), None)
    n, props, in_vertex = 0, [], False
    for ln in lines:
        tok = ln.split()
        if not tok:
            continue
        if tok[0] == "element":
            in_vertex = tok[1] == "vertex"
            if in_vertex:
                n = int(tok[2])
        elif tok[0] == "property" and in_vertex:
            if tok[1] == "list":
                raise ValueError(f"{path}: list properties on vertices are not supported")
            props.append((tok[2], _PLY_TYPES[tok[1]]))
    body = data[end + len(marker):]
    if fmt == "binary_little_endian":
        dt = np.dtype([(name, "<" + t) for name, t in props])
        arr = np.frombuffer(body, dtype=dt, count=n)
        return {name: np.ascontiguousarray(arr[name]) for name, _ in props}
    if fmt == "ascii":
        arr = np.loadtxt(body.decode("ascii").splitlines()[:n], dtype=np.float64, ndmin=2)
        return {name: arr[:, k].astype(t) for k, (name, t) in enumerate(props)}
    raise ValueError(f"{path}: unsupported PLY format {fmt}")


def morton_order(position: torch.Tensor, scale: torch.Tensor | None = None) -> torch.Tensor:
    """Stable argsort of (size class, 30-bit Morton code of the position) (device, torch).

    The render kernels stream Gaussians in slabs of 64 consecutive records and cull whole slabs
    by their union screen bound, so consecutive records should be spatially close.  This is the
    (camera-independent, load-time) stand-in for the reference's BVH build (scene.py:162-404).
    The sort is stable: exact duplicates keep their original relative order.

    When `scale` is given, the HUGE Gaussians (16 x the median size and more: the backgrounds of
    captured scenes) are ordered behind all others, Morton-ordered among themselves: a slab's
    screen bound is the union of its members', and one screen-filling member among 63 small ones
    voids it for every tile.  Measured (tools/cap_sweep.py, capture-like scene + 0.2 % / 1 % of
    40 x s0 Gaussians): 0.270 -> 0.241 ms and 0.583 -> 0.371 ms; a class per octave is worse (0.276 /
    0.439, and 0.185 -> 0.200 without any huge ones: it tears spatial neighbours apart).  Scenes
    without such outliers (c1..c5) keep the plain Morton order."""
    p = torch.nan_to_num(position.detach().to(torch.float64), nan=0.0, posinf=0.0, neginf=0.0)
    lo, hi = p.amin(dim=0), p.amax(dim=0)
    q = ((p - lo) / (hi - lo).clamp_min(1e-300) * 1023.0).round().clamp(0, 1023).to(torch.int64)

    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        return (v | (v << 2)) & 0x09249249

    code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    if scale is not None and HUGE_LAST:
        smax = torch.nan_to_num(scale.detach().to(torch.float64).abs().amax(dim=1), nan=0.0, posinf=0.0)
        med = smax.median().clamp_min(1e-300)
        code = code | ((smax >= 16.0 * med).to(torch.int64) << 30)
    return torch.argsort(code, stable=True)


class Scene:
    """scene.py:64-87.  max_num_node / balance_weight / leaf_prim are accepted for signature
    compatibility; they parameterise the reference's BVH, which this path does not build."""

    def __init__(self, max_num_node: int = 128, balance_weight: int = 1, leaf_prim=8) -> None:
        self.max_num_node = max_num_node
        self.balance_weight = balance_weight
        self.leaf_prim = leaf_prim
        self.bvh_field = None
        self.gaussian_field = None  # scene.py:83 allocates a shape-() field; filled by load
        self.num_gaussians = 0
        self.position = self.rotation = self.scale = self.color_opacity = self.sh = None
        self.version = 0            # bumped by commit(): invalidates cached projections
        self.order = None           # stream slot -> index into gaussian_field
        self.sort_spatially = True  # Morton-order the streamed copies (see morton_order)
        self._accel = None          # (geometry key, device buffer) of the arbitrary-ray structure

    # ---- construction ------------------------------------------------------------------
    @classmethod
    def from_arrays(cls, position, rotation, scale, color, opacity, sh=None, **kw) -> "Scene":
        """Bulk constructor: post-activation values, exactly what ``Gaussian.init`` takes
        (gaussian.py:59-84).  rotation is a unit quaternion xyzw; sh is (N, 15, 3) or None."""
        scene = cls(**kw)
        scene.set_arrays(position, rotation, scale, color, opacity, sh)
        return scene

    def set_arrays(self, position, rotation, scale, color, opacity, sh=None) -> None:
        dev = nat.device()

        def up(x, cols):
            a = np.ascontiguousarray(np.asarray(x, dtype=np.float32)).reshape((-1,) + cols)
            return torch.from_numpy(a).to(dev)

        field = GaussianField(up(position, (3,)).shape[0], device=dev)
        field.position, field.rotation = up(position, (3,)), up(rotation, (4,))
        field.scale, field.color = up(scale, (3,)), up(color, (3,))
        field.opacity = up(opacity, ())
        n = field.position.shape[0]
        for name, t in (("rotation", field.rotation), ("scale", field.scale),
                        ("color", field.color), ("opacity", field.opacity)):
            if t.shape[0] != n:
                raise ValueError(f"{name} has {t.shape[0]} rows, position has {n}")
        has_sh = sh is not None
        field.sh = up(sh, (15, 3)) if has_sh else torch.zeros((n, 15, 3), dtype=torch.float32, device=dev)
        field.n, field.shape = n, (n,)
        self.gaussian_field = field
        self.commit(has_sh)

    def commit(self, has_sh: bool | None = None) -> None:
        """(Re)build the device buffers the kernels stream from ``gaussian_field``: position,
        quaternion, scale (the covariance of gaussian.py:86-102 is formed from these on the
        device), colour and opacity packed to 16 B.  The streamed copies are stored in Morton order
        (``self.order`` maps a stream slot back to the ``gaussian_field`` index)."""
        f = self.gaussian_field
        n = f.n
        if n <= 0:
            raise ValueError("scene has no Gaussians")
        self.num_gaussians = n
        self.version += 1
        dev = f.position.device
        if self.position is not None:
            # a re-commit frees the streamed copies: frames still in flight on another stream (the
            # tile-parallel loop renders on its own) must have finished reading them
            torch.cuda.synchronize(dev)
        if self.sort_spatially and n > 64:
            self.order = morton_order(f.position, f.scale)
        else:
            self.order = torch.arange(n, device=dev)
        o = self.order
        self.position = f.position[o].contiguous()
        # quaternion + scale stay as they are: the kernels rebuild Sigma^-1 = R S^-2 R^T in fp64 per
        # frame (an fp32-rounded inverse covariance flips grazing hits of needle-like Gaussians)
        self.rotation = f.rotation[o].contiguous()
        self.scale = f.scale[o].contiguous()
        self.color_opacity = torch.cat([f.color, f.opacity[:, None]], dim=1)[o].contiguous()
        if has_sh is None:
            has_sh = bool(torch.any(f.sh != 0).item())
        self.sh = f.sh.reshape(n, 45)[o].contiguous() if has_sh else None

    def load_file(self, path: pathlib.Path, scale: float = 1, sh_layout: str = "inria"):
        """Load an INRIA-3DGS PLY (scene.py:89-131): wxyz -> xyzw (:103), normalised
        quaternions (:110-111), exp(scale) * scale (:112), sigmoid colour and opacity
        (:113-114), f_rest reshaped (-1, 3, 15) (:106-107).

        The kernel-side contract of the reference (scene.py:143-157) is that coefficient k of
        Gaussian i is an RGB triple.  INRIA 3DGS files store f_rest channel-major (15 red, 15 green,
        15 blue), which the reference reshapes to (-1, 3, 15) and hands to a (N, 15) vec3 field
        without a transpose (scene.py:106-107, 122, 127); what Taichi makes of that shape cannot be
        checked here.  ``sh_layout="inria"`` (default) transposes to (15, 3), the
        reading under which the colours are the ones the file's author trained;
        ``sh_layout="reference_raw"`` reinterprets the 45 values in memory order as (15, 3), what a
        flat copy into the reference's field would give."""
        if sh_layout not in ("inria", "reference_raw"):
            raise ValueError("sh_layout must be 'inria' or 'reference_raw'")
        p = read_ply(pathlib.Path(path))
        num_points = len(p["x"])
        logger.info(f"Point cloud loaded from {path} with {num_points} points.")
        f32 = np.float32
        positions = np.stack([p["x"], p["y"], p["z"]], axis=-1).astype(f32)
        rotations = np.stack([p["rot_1"], p["rot_2"], p["rot_3"], p["rot_0"]], axis=-1).astype(f32)
        scales = np.stack([p["scale_0"], p["scale_1"], p["scale_2"]], axis=-1).astype(f32)
        colors = np.stack([p["f_dc_0"], p["f_dc_1"], p["f_dc_2"]], axis=-1).astype(f32)
        sh = None
        if all(f"f_rest_{i}" in p for i in range(45)):
            sh = np.stack([p[f"f_rest_{i}"] for i in range(45)], axis=-1).astype(f32)
            if sh_layout == "inria":
                sh = np.ascontiguousarray(sh.reshape(-1, 3, 15).transpose(0, 2, 1))
            else:
                sh = np.ascontiguousarray(sh.reshape(-1, 15, 3))
        opacities = p["opacity"].astype(f32)
        rotations = rotations / np.linalg.norm(rotations, axis=-1)[:, np.newaxis]
        scales = (np.exp(scales) * f32(scale)).astype(f32)
        colors = sigmoid(colors)
        opacities = sigmoid(opacities)
        self.set_arrays(positions, rotations, scales, colors, opacities, sh)
        logger.info("Gaussian field built.")

    # ---- queries -----------------------------------------------------------------------
    def native(self) -> nat.RtgsScene:
        if self.position is None:
            raise RuntimeError("scene is empty: call load_file() or set_arrays() first")
        s = nat.RtgsScene()
        s.n = self.num_gaussians
        s.position = nat.ptr(self.position)
        s.rotation = nat.ptr(self.rotation)
        s.scale = nat.ptr(self.scale)
        s.color_opacity = nat.ptr(self.color_opacity)
        s.sh = nat.ptr(self.sh)
        return s

    def geometry_key(self) -> tuple:
        """Changes whenever the streamed geometry may have: ``commit()`` bumps ``version``; in-place
        edits of the streamed tensors (a supported contract: ``scene.position.mul_(2)``) bump their
        torch ``_version``.  Everything cached from the geometry (projections, ``accel``) is keyed on it."""
        return (self.version, self.position._version, self.rotation._version, self.scale._version)

    def content_key(self) -> tuple:
        """``geometry_key`` plus the shading tensors (colour / opacity, SH): changes whenever anything a
        rendered pixel depends on may have.  What deferred per-pixel state is keyed on."""
        return self.geometry_key() + (self.color_opacity._version, -1 if self.sh is None else self.sh._version)

    def accel(self) -> torch.Tensor:
        """The structure that stands in for the reference's BVH on arbitrary-ray queries
        (scene.py:162-404 builds it in ``load_file``; here it is built on first use, after every
        ``commit()`` and after an in-place edit of the streamed position / rotation / scale):
        fp64 inverse covariances + world-space slab boxes, ``rtgs_accel_build``."""
        key = self.geometry_key()
        if self._accel is None or self._accel[0] != key:
            need = nat.lib().rtgs_accel_bytes(self.num_gaussians)
            buf = torch.empty(need, dtype=torch.uint8, device=self.position.device)
            nat.check(nat.lib().rtgs_accel_build(self.native(), nat.ptr(buf), buf.numel(), nat.stream()),
                      "rtgs_accel_build")
            self._accel = (key, buf)
        return self._accel[1]

    def hit_batch(self, rays, brute_force: bool = False):
        """Closest hit of every ray of a ``RayField`` / (..., 8) device tensor
        (scene.py:406-450).  Returns (gaussian_idx int32 (...), intersections float32 (..., 2)).
        ``brute_force=True`` tests every Gaussian against every ray (same answers, no structure)."""
        data = rays.data if isinstance(rays, RayField) else rays
        data = nat.require_device(data, "rays")
        shape = data.shape[:-1]
        n = data.numel() // nat.RAY_FLOATS
        idx = torch.empty(shape, dtype=torch.int32, device=data.device)
        t = torch.empty(shape + (2,), dtype=torch.float32, device=data.device)
        if n:
            scene = self.native()
            acc = None if brute_force else self.accel()
            nat.check(nat.lib().rtgs_closest_hit_accel(scene, nat.ptr(acc), 0 if acc is None else acc.numel(), n,
                                                       nat.ptr(data), nat.ptr(idx), nat.ptr(t), nat.stream()),
                      "rtgs_closest_hit")
            # stream slot -> gaussian_field index
            idx = torch.where(idx >= 0, self.order[idx.clamp_min(0).long()].to(torch.int32), idx)
        return idx, t

    def hit(self, ray: Ray) -> SceneHit:
        """Ray cast against the scene: the Gaussian with the smallest entry distance t1 with
        ray.start < t1 < ray.end (scene.py:433-437)."""
        data = torch.from_numpy(ray.as_array()[None, :]).to(nat.device())
        idx, t = self.hit_batch(data)
        return SceneHit(int(idx[0]), t[0].cpu().numpy(), 0)
