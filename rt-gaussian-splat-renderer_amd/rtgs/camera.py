"""Pinhole camera -- host mirror of the reference's src/rtgs/camera.py."""

from __future__ import annotations

import numpy as np

from . import _native as nat
from .ray import Ray
from .utils import quaternion as quat
from .utils.types import vec2, vec2i, vec3, vec4


class Camera:
    """camera.py:8-29.  ``cam_ray_field`` is a (W, H) ``RayField`` indexed [i, j] with i = x,
    j = y and (0, 0) the bottom-left pixel; the sensor size equals buf_size (camera.py:26)."""

    def __init__(self, position, rotation, buf_size, focal_length):
        self.position = vec3(position)
        self.rotation = vec4(rotation)
        self.buf_size = vec2i(buf_size)
        self.censor_size = vec2i(buf_size)   # sic: the reference's spelling, camera.py:26
        self.focal_length = vec2(focal_length)
        if self.buf_size.x <= 0 or self.buf_size.y <= 0:
            raise ValueError("buf_size must be positive")
        # zero-initialised like the reference's field (camera.py:29): NOT camera rays until
        # generate_ray_field fills it, so a sample_step before that traces the stored (null) rays
        self.cam_ray_field = Ray.field(shape=(self.buf_size.x, self.buf_size.y))
        # pose the ray field was last generated for: what sample_step traces
        self.field_position = vec3(position)
        self.field_rotation = vec4(rotation)
        self.ray_field_generation = 0    # counts generate_ray_field() calls (not the tracer's deferred regenerations)

    def native(self, position=None, rotation=None, device_pose=None) -> nat.RtgsCamera:
        return nat.make_camera(self.position if position is None else position,
                               self.rotation if rotation is None else rotation,
                               self.buf_size.x, self.buf_size.y, self.focal_length.x,
                               self.focal_length.y, device_pose)

    def generate_ray(self, position, rotation, uv) -> Ray:
        """One camera ray through screen coordinate uv in [0,1]^2, (0,0) bottom-left
        (camera.py:31-55); the rotation runs on the device (``quat.rot_vec3``)."""
        uv = np.asarray(uv, dtype=np.float32)
        sensor = self.censor_size.astype(np.float32)
        pxy = (sensor * uv - np.float32(0.5) * sensor) / self.focal_length
        d = np.array([pxy[0], pxy[1], -1.0], dtype=np.float32)
        d = d / np.sqrt(np.sum(d * d, dtype=np.float32))
        dw = quat.rot_vec3(np.asarray(rotation, dtype=np.float32), d).cpu().numpy()
        ray = Ray()
        ray.init(position, dw)
        return ray

    def generate_ray_field(self, position, rotation) -> None:
        """Fill ``cam_ray_field`` with one ray per pixel centre (camera.py:57-71).  A deferred update
        that also owes per-pixel STATE (``RayTracer.sample_all``: the attenuation buffer lives in the
        tracer, not in this field) settles that part first (``RayField.supersede``) -- regenerating the
        rays supersedes the ``.start`` column, not the attenuation the eager calls would have left."""
        self.cam_ray_field.supersede()
        self.ray_field_generation += 1      # (RayTracer: a listed sample in progress no longer matches the field's .start)
        self._generate(position, rotation)
        self.cam_ray_field.defer(None, False)      # up to date; supersedes any deferred update
        self.field_position = vec3(position)
        self.field_rotation = vec4(rotation)

    def _generate(self, position, rotation) -> None:
        cam = self.native(position, rotation)
        nat.check(nat.lib().rtgs_generate_rays(cam, nat.ptr(self.cam_ray_field.storage), nat.stream()),
                  "rtgs_generate_rays")

    def defer_ray_field(self, position, rotation, update, state_deferred: bool, superseded=None) -> None:
        """What ``RayTracer`` calls instead of ``generate_ray_field`` at the start of a sample: the
        field *logically* holds the rays of this pose from now on, but its 32 B per pixel are only
        rewritten (by ``update``) when somebody reads or edits it (``RayField`` docstring)."""
        self.cam_ray_field.defer(update, state_deferred, superseded)
        self.field_position = vec3(position)
        self.field_rotation = vec4(rotation)
