"""Ray-traced Gaussian-splat renderer -- host mirror of the reference's src/rtgs/ray_tracer.py.

The reference peels one Gaussian per ``sample_step`` launch and needs ``depth`` launches per
image, each re-walking a BVH (ray_tracer.py:39-54, 79-104).  The HIP kernel behind
``rtgs_trace`` finds the ``depth`` nearest entry distances of every ray in ONE pass and
composites them front to back; ``sample_step`` is the same kernel asked for a single peel and
resuming from the per-pixel state (``cam_ray_field.start``, ``attenuation_buf``), so the
progressive state machine of the reference's viewer loop (__main__.py:253-257) is preserved.
"""

from __future__ import annotations

import os

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
        self._attenuation = torch.zeros((w, h), dtype=torch.float32, device=dev)
        self.num_steps = 0
        self.num_samples = 0
        self.disp_buf = torch.zeros((w, h, 3), dtype=torch.float32, device=dev)
        self._workspace = None
        self._projected_for = None   # (scene version, camera pose) the workspace records belong to
        self._orders = {}            # (tile_first, tile_step) -> device int32 launch order
        # RTGS_TILE_ORDER as in parallel.HipBackend: 0 = tile order, centre, wedge (default)
        self._order_mode = os.environ.get("RTGS_TILE_ORDER", "wedge")
        self._replay_owed = False    # the ray field's pending update is sample_all's replay (it owes the attenuation too)
        self._att_cleared = False    # clear_attenuation() ran while that replay was still owed
        # peel lists (rtgs_trace_lists / rtgs_peel): the sample() pattern from one traversal.  RTGS_PEEL_LISTS=0:
        # every sample() re-traverses (the K = 1 resumed launch), as up to round 3
        self._use_lists = os.environ.get("RTGS_PEEL_LISTS", "1") != "0"
        self._lists = None           # device buffer, rtgs_peel_list_bytes(W, H)
        self._lists_for = None       # (scene, geometry, pose, focal) the lists were traversed for
        self._list_peels = None      # peels of the CURRENT sample composited from the lists; None = not in step with them
        self._state_peels = 0        # peels the ray field's deferred .start update has to account for (rtgs_peel_start)
        self._sample_pose = None     # (position, rotation) the listed sample in progress was started from
        self._list_gen = -1          # camera.ray_field_generation when the last listed peel was composited

    @property
    def attenuation_buf(self) -> torch.Tensor:
        """ray_tracer.py:34.  After ``sample_all`` the per-pixel state (this buffer and
        ``cam_ray_field.start``) is only written when somebody looks (``RayField``: lazy contents)."""
        f = self.camera.cam_ray_field
        if f.pending and f.state_deferred:
            f.materialise()
        return self._attenuation

    # ---- plumbing ------------------------------------------------------------------------
    def _ws(self) -> torch.Tensor:
        need = nat.lib().rtgs_workspace_bytes_frame(self.scene.num_gaussians, self.buf_size.x, self.buf_size.y)
        if self._workspace is None or self._workspace.numel() < need:
            self._workspace = torch.empty(need, dtype=torch.uint8, device=self.sample_buf.device)
        return self._workspace

    def _tile_order(self, tile_first: int, tile_step: int):
        """The XCD-wedge launch order (``rtgs_launch_order``; scheduling only), cached per shard."""
        if self._order_mode == "0":
            return None
        key = (tile_first, tile_step)
        if key not in self._orders:
            from .parallel import cost_sorted_order
            o = cost_sorted_order(self.buf_size.x, self.buf_size.y, nat.tile_size(), tile_first, tile_step,
                                  "centre" if self._order_mode == "centre" else "wedge")
            self._orders[key] = torch.from_numpy(o).to(self.sample_buf.device) if o.size else None
        return self._orders[key]

    def trace(self, depth: int, *, flags: int = 0, t_cutoff: float = 0.0, tile_first: int = 0,
              tile_step: int = 1, sample_buf: torch.Tensor | None = None, position=None,
              rotation=None, projected: bool = False, state: bool = True) -> None:
        """One ``rtgs_trace`` call on the current stream (see include/rtgs_hip.h).  ``state=False``
        leaves ``attenuation_buf`` and ``cam_ray_field.start`` alone (NULL pointers: the kernel then
        neither reads nor writes them); a depth above ``RTGS_MAX_DEPTH`` needs them (the library
        chains launches through that state)."""
        ws = self._ws()
        params = nat.RtgsTraceParams(int(depth), float(t_cutoff), int(tile_first), int(tile_step),
                                     int(flags), nat.ptr(self._tile_order(int(tile_first), int(tile_step))))
        # the per-frame projection only depends on (scene, camera pose): the 16 single-peel calls of
        # one progressive sample share it
        pos = self.camera.position if position is None else position
        rot = self.camera.rotation if rotation is None else rotation
        sc = self.scene
        pstate = (id(sc), sc.geometry_key(), ws.data_ptr(),
                  tuple(float(v) for v in pos), tuple(float(v) for v in rot),
                  tuple(float(v) for v in self.camera.focal_length))
        if not projected and pstate == self._projected_for:
            projected = True
        self._projected_for = pstate
        out = self.sample_buf if sample_buf is None else nat.require_device(sample_buf, "sample_buf")
        att_ptr = start_ptr = None
        if state:
            att_ptr = nat.ptr(self._attenuation)
            # &cam_ray_field[0,0].start, stride 8 floats (raw storage: the kernel owns that column)
            start_ptr = self.camera.cam_ray_field.storage.data_ptr() + 6 * 4
        nat.check(nat.lib().rtgs_trace(self.scene.native(), self.camera.native(position, rotation),
                                       params, nat.ptr(out), att_ptr, start_ptr, nat.RAY_FLOATS,
                                       nat.ptr(ws), ws.numel(), 1 if projected else 0, nat.stream()),
                  "rtgs_trace")

    def _pose(self):
        return tuple(float(v) for v in self.camera.position), tuple(float(v) for v in self.camera.rotation)

    def _regenerate_keeping_start(self, pos, rot) -> None:
        """Deferred update of a ray field whose ``.start`` column is live (the peel kernels wrote
        it): origin / direction / end of the pose the sample started from, ``.start`` untouched."""
        f = self.camera.cam_ray_field
        keep = f.storage[..., 6].clone()
        self.camera._generate(pos, rot)
        f.storage[..., 6] = keep

    def _state_key(self):
        """Everything besides the pose that the per-pixel state of a sample depends on."""
        return (id(self.scene), self.scene.content_key(), tuple(float(v) for v in self.camera.focal_length))

    def _replay_state(self, pos, rot, depth: int, t_cutoff: float, key, rays_superseded: bool = False) -> None:
        """Deferred update after ``sample_all``: generated rays of that pose, then the per-pixel
        state (``.start``, attenuation) the ``depth`` peels leave, by tracing once more into a
        scratch image.  Only the scene and focal length the sample was taken with can give that
        state back: if either changed in between, the state of that sample is gone and saying so
        beats handing out the state of a frame that was never rendered.  `rays_superseded`: the rays
        are being regenerated anyway (``Camera.generate_ray_field``), only the attenuation is still owed
        -- and not even that once ``clear_attenuation()`` has run."""
        self._replay_owed = False
        if rays_superseded and self._att_cleared:
            self._att_cleared = False
            return
        if key != self._state_key():
            self._replay_owed = True          # (the debt stands: RayField.materialise keeps the update pending)
            raise RuntimeError(
                "the scene (or focal length) changed after sample_all(): the per-pixel state of that sample "
                "(attenuation_buf, cam_ray_field.start) can no longer be reconstructed.  Read it before "
                "editing the scene, or take the sample with sample_all(..., lazy_state=False)")
        self.camera._generate(pos, rot)
        scratch = torch.empty_like(self.sample_buf)
        self.trace(depth, t_cutoff=t_cutoff, sample_buf=scratch, position=pos, rotation=rot)
        if self._att_cleared:          # the caller has cleared the attenuation since: ones are what it holds
            self._att_cleared = False
            self.clear_attenuation()

    # ---- peel lists -----------------------------------------------------------------------
    def _lists_key(self, pos, rot):
        return (id(self.scene), self.scene.geometry_key(), pos, rot, tuple(float(v) for v in self.camera.focal_length))

    def _ensure_lists(self, pos, rot) -> None:
        """Every ray's sorted distinct entries for this (geometry, pose), by ONE traversal
        (``rtgs_trace_lists``); kept across samples while neither changes -- the reference's samples of
        a still camera are identical (no jitter: ``random_vec3`` is unused), so is what they traverse."""
        key = self._lists_key(pos, rot)
        if self._lists_for == key:
            return
        lib = nat.lib()
        w, h = self.buf_size.x, self.buf_size.y
        need = lib.rtgs_peel_list_bytes(w, h)
        if self._lists is None or self._lists.numel() < need:
            self._lists = torch.empty(need, dtype=torch.uint8, device=self.sample_buf.device)
        ws = self._ws()
        pstate = (id(self.scene), self.scene.geometry_key(), ws.data_ptr(), pos, rot,
                  tuple(float(v) for v in self.camera.focal_length))
        projected = pstate == self._projected_for
        self._projected_for = pstate
        self._lists_for = None
        nat.check(lib.rtgs_trace_lists(self.scene.native(), self.camera.native(pos, rot), nat.ptr(self._tile_order(0, 1)),
                                       nat.ptr(self._lists), self._lists.numel(), nat.ptr(ws), ws.numel(),
                                       1 if projected else 0, nat.stream()), "rtgs_trace_lists")
        self._lists_for = key

    def _start_from_lists(self, pos, rot) -> None:
        """Deferred update of the ray field during / after a listed sample: generated rays of the
        sample's pose, ``.start`` as the peels composited so far leave it (``rtgs_peel_start``)."""
        self.camera._generate(pos, rot)
        f = self.camera.cam_ray_field
        nat.check(nat.lib().rtgs_peel_start(self.camera.native(pos, rot), nat.ptr(self._lists), self._lists.numel(),
                                            int(self._state_peels), f.storage.data_ptr() + 6 * 4, nat.RAY_FLOATS,
                                            nat.stream()), "rtgs_peel_start")

    # ---- the reference's API ------------------------------------------------------------
    def sample(self, depth: int):
        """Advance the current sample by one peel (ray_tracer.py:39-54).  The reference starts a
        sample with ``clear_attenuation()`` + ``generate_ray_field()`` (90 MB of writes at 1080p) and
        walks its BVH once per peel.  Here the first ``sample()`` for a (geometry, pose) traverses the
        scene ONCE and keeps every ray's sorted entries (``rtgs_trace_lists``); this and the following
        calls composite one listed entry each (``rtgs_peel``: a streaming kernel), for this sample and
        for the next ones while scene geometry and camera stand still.  The rays and ``.start`` are
        brought up to date only if somebody reads them (``RayField``: lazy contents).  Beyond 16 peels
        per sample, after an edit of the ray field or of the geometry mid-sample, or with
        RTGS_PEEL_LISTS=0, a peel is the K = 1 launch resuming from the per-pixel state."""
        first = self.num_steps == 0
        f = self.camera.cam_ray_field
        if first:
            pos, rot = self._pose()
            self._list_peels = None
            if self._use_lists:
                self._ensure_lists(pos, rot)
                self._list_peels = 0
                self._sample_pose = (pos, rot)
            else:
                self.camera.defer_ray_field(pos, rot, lambda: self._regenerate_keeping_start(pos, rot), False)
        k = self.num_steps
        # (a generate_ray_field() since the last listed peel put .start back to 0: the reference's next sample_step
        # peels the NEAREST entry again -- found by tests/test_gpu_state_machine.py, round 5)
        listed = (self._list_peels == k and k < nat.MAX_DEPTH and (first or not f.edited) and
                  (first or self._list_gen == self.camera.ray_field_generation) and
                  self._lists_for is not None and self._lists_for[1] == self.scene.geometry_key())
        if listed:
            pos, rot = self._sample_pose
            nat.check(nat.lib().rtgs_peel(self.scene.native(), self.camera.native(pos, rot), nat.ptr(self._lists),
                                          self._lists.numel(), k, nat.PEEL_FRESH if first else 0,
                                          nat.ptr(self.sample_buf), nat.ptr(self._attenuation), nat.stream()), "rtgs_peel")
            self._list_peels = self._state_peels = k + 1
            self._replay_owed = False
            # .start (and the rays) are owed again: what a reader sees is the state after k + 1 peels
            self.camera.defer_ray_field(pos, rot, lambda: self._start_from_lists(pos, rot), True)
            self._list_gen = self.camera.ray_field_generation
        else:
            self.sample_step(_fresh=first and self._list_peels is None)
        self.num_steps += 1
        if self.num_steps >= depth:
            self.num_steps = 0
            self.num_samples += 1

    def sample_step(self, _fresh: bool = False):
        """Composite the next-nearest Gaussian of every pixel (ray_tracer.py:79-104), resuming
        from ``cam_ray_field.start`` / ``attenuation_buf`` and accumulating into ``sample_buf``.

        A ray field that still holds what ``generate_ray_field`` wrote takes the screen-space
        kernel (``rtgs_trace``); one the caller has edited (``cam_ray_field[i, j] = ray``,
        ``.init()``, ``.mark_edited()``) is traced ray by ray as the reference does, whatever the
        rays are (``rtgs_sample_step_rays``)."""
        f = self.camera.cam_ray_field
        if f.pending and f.state_deferred:
            f.materialise()              # peeling on after a sample_all / listed peels: the state has to exist
        self._list_peels = None          # from here on the per-pixel state, not the lists, says where the sample is
        if f.edited:
            self.sample_step_rays()
            return
        self.trace(1, flags=nat.TRACE_ACCUMULATE | (0 if _fresh else nat.TRACE_RESUME),
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
                                                        nat.ptr(self.sample_buf), nat.ptr(self._attenuation),
                                                        nat.stream()), "rtgs_sample_step_rays")

    def sample_all(self, depth: int, t_cutoff: float = 0.0, lazy_state: bool = True):
        """Extension: one WHOLE sample (``depth`` peels) in a single call -- the result of
        calling ``sample(depth)`` ``depth`` times from num_steps == 0.  Any depth: above
        ``RTGS_MAX_DEPTH`` (16) the library chains ceil(depth / 16) launches through the per-pixel
        state (the reference's ``-d`` is unbounded, __main__.py:66-71).  Up to 16 the state
        (``attenuation_buf``, ``cam_ray_field``) is not written at all unless somebody reads it
        afterwards: the frame costs what the tile-parallel path's does.  The deferred state is that of
        THIS scene and focal length: editing either before the state is read makes the read raise
        (``lazy_state=False`` writes it with the sample, 90 MB more traffic per 1080p frame)."""
        if self.num_steps != 0:
            raise RuntimeError("sample_all() needs num_steps == 0 (finish the partial sample first)")
        depth = int(depth)
        pos, rot = self._pose()
        self._att_cleared = self._replay_owed = False
        self._list_peels = None              # the per-pixel state is this sample's from here on, not the lists'
        if depth <= nat.MAX_DEPTH and lazy_state:
            self.trace(depth, flags=nat.TRACE_ACCUMULATE, t_cutoff=t_cutoff, state=False)
            key = self._state_key()
            self._replay_owed = True
            self.camera.defer_ray_field(pos, rot, lambda: self._replay_state(pos, rot, depth, t_cutoff, key), True,
                                        lambda: self._replay_state(pos, rot, depth, t_cutoff, key, True))
        else:
            self.trace(depth, flags=nat.TRACE_ACCUMULATE, t_cutoff=t_cutoff)
            self.camera.defer_ray_field(pos, rot, lambda: self._regenerate_keeping_start(pos, rot), False)
        self.num_samples += 1

    def clear_sample(self):
        """ray_tracer.py:56-60."""
        nat.check(nat.lib().rtgs_fill(nat.ptr(self.sample_buf), self.sample_buf.numel(), 0.0,
                                      nat.stream()), "rtgs_fill")

    def clear_attenuation(self):
        """ray_tracer.py:62-66.  While ``sample_all``'s per-pixel state is still owed, the attenuation
        part of that debt is cancelled: the replay ends by filling ones again."""
        if self.camera.cam_ray_field.pending and self._replay_owed:
            self._att_cleared = True
        nat.check(nat.lib().rtgs_fill(nat.ptr(self._attenuation), self._attenuation.numel(),
                                      1.0, nat.stream()), "rtgs_fill")

    def generate_disp_buffer(self, num_samples: int, num_steps: int, num_depth: int):
        """disp = sample / (num_samples + num_steps / num_depth)  (ray_tracer.py:68-77)."""
        w, h = self.buf_size.x, self.buf_size.y
        nat.check(nat.lib().rtgs_disp_buffer(w * h, nat.ptr(self.sample_buf), int(num_samples),
                                             int(num_steps), int(num_depth), nat.ptr(self.disp_buf),
                                             nat.stream()), "rtgs_disp_buffer")
