"""Ray-traced Gaussian-splat renderer -- host mirror of the reference's src/rtgs/ray_tracer.py.

The reference peels one Gaussian per ``sample_step`` launch and needs ``depth`` launches per
image, each re-walking a BVH (ray_tracer.py:39-54, 79-104).  The HIP kernel behind
``rtgs_trace`` finds the ``depth`` nearest entry distances of every ray in ONE pass and
composites them front to back; ``sample_step`` is the same kernel asked for a single peel and
resuming from the per-pixel state (``cam_ray_field.start``, ``attenuation_buf``), so the
progressive state machine of the reference's viewer loop (__main__.py:253-257) is preserved.
"""

from __future__ import annotations

import torch

from . import _native as nat
from .camera import Camera
from .scene import Scene
from .utils.types import vec2i


class RayTracer:
    def __init__(self, buf_size, scene: Scene, camera: Camera) -> None:
        self.scene = scene
        self.camera = camera
        self.buf_size = vec2i(buf_size)
        if tuple(self.buf_size) != tuple(camera.buf_size):
            raise ValueError("RayTracer buf_size must equal the camera's buf_size")
        dev = nat.device()
        w, h = self.buf_size.x, self.buf_size.y
        # ray_tracer.py:33-37
        self.sample_buf = torch.zeros((w, h, 3), dtype=torch.float32, device=dev)
        self.attenuation_buf = torch.zeros((w, h), dtype=torch.float32, device=dev)
        self.num_steps = 0
        self.num_samples = 0
        self.disp_buf = torch.zeros((w, h, 3), dtype=torch.float32, device=dev)
        self._workspace = None
        self._projected_for = None   # (scene version, camera pose) the workspace records belong to

    # ---- plumbing ------------------------------------------------------------------------
    def _ws(self) -> torch.Tensor:
        need = nat.lib().rtgs_workspace_bytes(self.scene.num_gaussians)
        if self._workspace is None or self._workspace.numel() < need:
            self._workspace = torch.empty(need, dtype=torch.uint8, device=self.sample_buf.device)
        return self._workspace

    def trace(self, depth: int, *, flags: int = 0, t_cutoff: float = 0.0, tile_first: int = 0,
              tile_step: int = 1, sample_buf: torch.Tensor | None = None, position=None,
              rotation=None, projected: bool = False) -> None:
        """One ``rtgs_trace`` launch on the current stream (see include/rtgs_hip.h)."""
        ws = self._ws()
        params = nat.RtgsTraceParams(int(depth), float(t_cutoff), int(tile_first), int(tile_step),
                                     int(flags), None)
        # the per-frame projection only depends on (scene, camera pose): the 16 single-peel calls of
        # one progressive sample share it
        pos = self.camera.position if position is None else position
        rot = self.camera.rotation if rotation is None else rotation
        sc = self.scene
        state = (id(sc), sc.version, sc.position._version, sc.rotation._version, sc.scale._version, ws.data_ptr(),
                 tuple(float(v) for v in pos), tuple(float(v) for v in rot),
                 tuple(float(v) for v in self.camera.focal_length))
        if not projected and state == self._projected_for:
            projected = True
        self._projected_for = state
        out = self.sample_buf if sample_buf is None else nat.require_device(sample_buf, "sample_buf")
        rays = self.camera.cam_ray_field.data
        start_ptr = rays.data_ptr() + 6 * 4  # &cam_ray_field[0,0].start, stride 8 floats
        nat.check(nat.lib().rtgs_trace(self.scene.native(), self.camera.native(position, rotation),
                                       params, nat.ptr(out), nat.ptr(self.attenuation_buf),
                                       start_ptr, nat.RAY_FLOATS, nat.ptr(ws), ws.numel(),
                                       1 if projected else 0, nat.stream()), "rtgs_trace")

    # ---- the reference's API ------------------------------------------------------------
    def sample(self, depth: int):
        """Advance the current sample by one peel (ray_tracer.py:39-54)."""
        if self.num_steps == 0:
            self.clear_attenuation()
            self.camera.generate_ray_field(self.camera.position, self.camera.rotation)
        self.sample_step()
        self.num_steps += 1
        if self.num_steps >= depth:
            self.num_steps = 0
            self.num_samples += 1

    def sample_step(self):
        """Composite the next-nearest Gaussian of every pixel (ray_tracer.py:79-104), resuming
        from ``cam_ray_field.start`` / ``attenuation_buf`` and accumulating into ``sample_buf``.

        A ray field that still holds what ``generate_ray_field`` wrote takes the screen-space
        kernel (``rtgs_trace``); one the caller has edited (``cam_ray_field[i, j] = ray``,
        ``.init()``, ``.mark_edited()``) is traced ray by ray as the reference does, whatever the
        rays are (``rtgs_sample_step_rays``)."""
        if self.camera.cam_ray_field.edited:
            self.sample_step_rays()
            return
        self.trace(1, flags=nat.TRACE_RESUME | nat.TRACE_ACCUMULATE,
                   position=self.camera.field_position, rotation=self.camera.field_rotation)

    def sample_step_rays(self, brute_force: bool = False):
        """The literal per-ray form of ``sample_step``: ``Scene.hit`` of every ray of
        ``cam_ray_field`` exactly as stored (origin, direction, start, end), through the scene's
        slab-box structure (``Scene.accel``) or, ``brute_force=True``, against every Gaussian."""
        rays = self.camera.cam_ray_field.data
        n = self.buf_size.x * self.buf_size.y
        acc = None if brute_force else self.scene.accel()
        nat.check(nat.lib().rtgs_sample_step_rays_accel(self.scene.native(), nat.ptr(acc),
                                                        0 if acc is None else acc.numel(), n, nat.ptr(rays),
                                                        nat.ptr(self.sample_buf), nat.ptr(self.attenuation_buf),
                                                        nat.stream()), "rtgs_sample_step_rays")

    def sample_all(self, depth: int, t_cutoff: float = 0.0):
        """Extension: one WHOLE sample (``depth`` peels) in a single launch -- the result of
        calling ``sample(depth)`` ``depth`` times from num_steps == 0."""
        if self.num_steps != 0:
            raise RuntimeError("sample_all() needs num_steps == 0 (finish the partial sample first)")
        self.camera.generate_ray_field(self.camera.position, self.camera.rotation)
        self.trace(depth, flags=nat.TRACE_ACCUMULATE, t_cutoff=t_cutoff)
        self.num_samples += 1

    def clear_sample(self):
        """ray_tracer.py:56-60."""
        nat.check(nat.lib().rtgs_fill(nat.ptr(self.sample_buf), self.sample_buf.numel(), 0.0,
                                      nat.stream()), "rtgs_fill")

    def clear_attenuation(self):
        """ray_tracer.py:62-66."""
        nat.check(nat.lib().rtgs_fill(nat.ptr(self.attenuation_buf), self.attenuation_buf.numel(),
                                      1.0, nat.stream()), "rtgs_fill")

    def generate_disp_buffer(self, num_samples: int, num_steps: int, num_depth: int):
        """disp = sample / (num_samples + num_steps / num_depth)  (ray_tracer.py:68-77)."""
        w, h = self.buf_size.x, self.buf_size.y
        nat.check(nat.lib().rtgs_disp_buffer(w * h, nat.ptr(self.sample_buf), int(num_samples),
                                             int(num_steps), int(num_depth), nat.ptr(self.disp_buf),
                                             nat.stream()), "rtgs_disp_buffer")
