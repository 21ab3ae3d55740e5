"""Ray record and ray fields -- host mirror of the reference's src/rtgs/ray.py.

A ``Ray`` is the reference's 32-byte struct {origin vec3, direction vec3, start f32, end f32}
(ray.py:4-18).  ``Ray.field(shape)`` returns a ``RayField``: a zero-initialised torch tensor of
shape ``shape + (8,)`` on the HIP device, laid out exactly like the reference's AoS Taichi
field, which is what ``Camera.cam_ray_field`` is (camera.py:29).
"""

from __future__ import annotations

import math

import numpy as np
import torch

from . import _native as nat
from .utils.types import vec3


class Ray:
    """ray.py:4-18.  ``Ray()`` is zero-initialised, like a Taichi dataclass instance."""

    __slots__ = ("origin", "direction", "start", "end")

    def __init__(self, origin=(0, 0, 0), direction=(0, 0, 0), start=0.0, end=0.0):
        self.origin = vec3(origin)
        self.direction = vec3(direction)
        self.start = float(np.float32(start))
        self.end = float(np.float32(end))

    def init(self, origin=(0, 0, 0), direction=(0, 1, 0), start=0, end=math.inf):
        """ray.py:20-41: defaults origin 0, direction +y, start 0, end inf."""
        self.origin = vec3(origin)
        self.direction = vec3(direction)
        self.start = float(np.float32(start))
        self.end = float(np.float32(end))

    def as_array(self) -> np.ndarray:
        return np.concatenate([self.origin, self.direction, [self.start, self.end]]).astype(np.float32)

    @classmethod
    def from_array(cls, a) -> "Ray":
        a = np.asarray(a, dtype=np.float32).reshape(8)
        return cls(a[0:3], a[3:6], a[6], a[7])

    def get(self, t):
        """ray.py:43-52: o + t d, evaluated by the ``rtgs_ray_get`` kernel."""
        dev = nat.device()
        rays = torch.from_numpy(self.as_array()[None, :]).to(dev)
        tt = torch.tensor([t], dtype=torch.float32, device=dev)
        out = torch.empty((1, 3), dtype=torch.float32, device=dev)
        nat.check(nat.lib().rtgs_ray_get(1, nat.ptr(rays), nat.ptr(tt), nat.ptr(out), nat.stream()),
                  "rtgs_ray_get")
        return vec3(out[0].cpu().numpy())

    @staticmethod
    def field(shape) -> "RayField":
        return RayField(shape)

    def __repr__(self):
        return (f"Ray(origin={self.origin.tolist()}, direction={self.direction.tolist()}, "
                f"start={self.start}, end={self.end})")


class RayField:
    """A (shape..., 8) float32 device tensor of rays; element [I] is a ``Ray``.

    ``edited`` tells whether the field may differ from what ``Camera.generate_ray_field`` last
    wrote: element assignment, ``init`` and ``mark_edited()`` set it, and so does any in-place
    torch write through ``.data`` or the component views (``field.start[...] = x``), which bumps
    the tensor's version counter.  A new field is edited: zeros are not camera rays.  The HIP
    kernels write through raw pointers and do not count.  ``RayTracer.sample_step`` picks its
    kernel by it.

    **Lazy contents.**  The screen-space kernels generate camera rays in registers and never read
    the origin / direction of this field, so ``RayTracer`` does not rewrite its 32 B per pixel for
    every sample (camera.py:57-71 does, 66 MB at 1080p): it *defers* the update (``defer``) and the
    field brings itself up to date the first time anybody looks -- ``.data``, a component view,
    element access, ``to_numpy``, ``get``.  What a reader sees is exactly what the eager calls would
    have left.  Code that only needs the storage (the kernels' raw pointers) uses ``storage``."""

    def __init__(self, shape, data: torch.Tensor | None = None):
        self.shape = (shape,) if isinstance(shape, int) else tuple(int(s) for s in shape)
        if data is None:
            data = torch.zeros(self.shape + (nat.RAY_FLOATS,), dtype=torch.float32,
                               device=nat.device())
        self._data = nat.require_device(data, "ray field")
        self._edited = True     # not (yet) a generated camera ray field
        self._clean_version = self._data._version
        self._pending = None    # deferred update: callable that brings the tensor up to date
        self._pending_superseded = None   # what is left of that update when the rays are regenerated anyway
        self.state_deferred = False   # the deferred update also owes the .start column (and the tracer's attenuation)

    # ---- deferred updates -----------------------------------------------------------------
    @property
    def storage(self) -> torch.Tensor:
        """The tensor WITHOUT bringing it up to date (raw-pointer users: the HIP kernels)."""
        return self._data

    @property
    def data(self) -> torch.Tensor:
        self.materialise()
        return self._data

    @data.setter
    def data(self, value: torch.Tensor) -> None:
        self._pending, self._pending_superseded, self.state_deferred = None, None, False
        self._data = nat.require_device(value, "ray field")
        self.mark_edited(True)

    @property
    def pending(self) -> bool:
        return self._pending is not None

    def defer(self, update, state_deferred: bool, superseded=None) -> None:
        """Replace whatever update was pending by `update` (a whole-field regeneration supersedes
        any earlier one) and mark the field clean: its *logical* contents are generated rays.
        `superseded`: what remains to be done of `update` if the rays are regenerated before anybody
        looked (state that lives OUTSIDE this field: the tracer's attenuation buffer)."""
        self._pending, self.state_deferred = update, bool(state_deferred)
        self._pending_superseded = superseded if update is not None else None
        self.mark_edited(False)

    def materialise(self) -> None:
        if self._pending is not None:
            saved = (self._pending, self._pending_superseded, self.state_deferred)
            self._pending, self._pending_superseded, self.state_deferred = None, None, False
            try:
                saved[0]()
            except BaseException:
                # the debt stands (a reader keeps being told) until a new sample or regeneration replaces it
                self._pending, self._pending_superseded, self.state_deferred = saved
                raise
            self.mark_edited(False)   # the update wrote through torch / raw pointers: still generated rays

    def supersede(self) -> None:
        """The rays are about to be regenerated: drop the pending update, after settling the part of
        it that does not live in this field (``defer``'s `superseded`)."""
        rest = self._pending_superseded
        if self._pending is not None and rest is not None:
            rest()
        self._pending, self._pending_superseded, self.state_deferred = None, None, False

    @property
    def edited(self) -> bool:
        return self._edited or self._data._version != self._clean_version

    @edited.setter
    def edited(self, value: bool) -> None:
        self.mark_edited(value)

    def mark_edited(self, edited: bool = True) -> None:
        if edited:
            self.materialise()   # an edit applies on top of the up-to-date contents
        self._edited = bool(edited)
        self._clean_version = self._data._version

    def init(self, origin=(0, 0, 0), direction=(0, 1, 0), start=0, end=math.inf):
        """Run ``Ray.init`` (ray.py:20-41) on every element."""
        proto = Ray()
        proto.init(origin, direction, start, end)
        self.data[...] = torch.from_numpy(proto.as_array()).to(self.data.device)
        self.edited = True

    def get(self, t) -> torch.Tensor:
        """``Ray.get`` (ray.py:43-52) elementwise; t broadcasts against the field shape."""
        n = math.prod(self.shape)
        tt = torch.as_tensor(t, dtype=torch.float32, device=self.data.device)
        tt = tt.expand(self.shape).contiguous()
        out = torch.empty(self.shape + (3,), dtype=torch.float32, device=self.data.device)
        if n:
            nat.check(nat.lib().rtgs_ray_get(n, nat.ptr(self.data), nat.ptr(tt), nat.ptr(out),
                                             nat.stream()), "rtgs_ray_get")
        return out

    # views of the AoS components
    origin = property(lambda s: s.data[..., 0:3])
    direction = property(lambda s: s.data[..., 3:6])
    start = property(lambda s: s.data[..., 6])
    end = property(lambda s: s.data[..., 7])

    def __getitem__(self, idx) -> Ray:
        return Ray.from_array(self.data[idx].cpu().numpy())

    def __setitem__(self, idx, ray: Ray):
        self.data[idx] = torch.from_numpy(ray.as_array()).to(self.data.device)
        self.edited = True

    def to_numpy(self) -> np.ndarray:
        return self.data.cpu().numpy()


def new_ray(origin=(0, 0, 0), direction=(0, 1, 0), start=0, end=math.inf) -> Ray:
    """Python-scope constructor (ray.py:55-68)."""
    return Ray(origin, direction, start, end)
