"""Tile-parallel rendering: one process per GPU, macro-tiles interleaved over ranks, RCCL exchange.

The reference is single-process / single-device (no collective anywhere).  Pixels
are independent, so the frame shards naturally: the T x T macro-tiles (T = rtgs_tile_size())
are dealt round-robin (tile t -> rank t % world), every rank holds the whole scene (one broadcast
at load, Gaussians are replicated) and renders its tiles into a compact [tiles_per_rank, T, T, 3]
buffer, and ONE exchange step lands the shards of a frame on its destination rank, where
``rtgs_untile`` de-interleaves them into the (W, H, 3) framebuffer.  Every rank keeps F frames in
flight (``TileParallelRenderer``, default 2): frame f is traced on render stream f mod F while the
exchange of the frames before it runs on the caller's stream.

Where a frame lands (``landing``):
  "root"    every frame on rank ``dst``: a ``torch.distributed`` gather (backend "nccl" == RCCL over
            xGMI) of W*H*12/world bytes per rank.  What a single display loop wants
            (/root/reference's consumer is one, __main__.py:253-257) -- and a bound: the root receives
            (world-1)/world of every frame, one shard per link, and untiles every frame.
  "rotate"  (round 5) the roots rotate: frame f on rank f mod world when a batch is world frames; in
            general a batch of B = k * world consecutive frames travels in ONE ``all_to_all_single`` and
            rank d receives, from everybody, the shards of the batch's frames [d k, (d + 1) k) -- world
            gathers with world different roots in one collective, every xGMI link carrying 1/world of
            what the single root's links carry, every rank untiling one frame in world.  For consumers
            that can take frames from any rank (an encoder per GPU, a throughput benchmark).

``render_frames(poses)`` queues any number of frames from an array of poses: per batch one library
call on the render side (``rtgs_shard_frames``: the batch's frames in runs of up to 8 per LAUNCH --
one k_project / k_bin / k_trace with grid.y = frames, so that small shards fill the chip together --
dealt over the F render streams), one collective, one landing call (``rtgs_frames_land``).

Round 4: the deal need not be equal.  ``Partition`` names every rank's tiles explicitly (``RtgsTraceParams.tile_list``),
``Partition.weighted`` gives every rank its own share, the exchange is then point to point (each shard's own bytes) and
the destination scatters through a tile -> (rank, slot) map (``rtgs_frames_land_map``); ``partition="auto"`` measures,
on the machine it runs on, the share the destination should keep (``calibrated_shares``).
"""

from __future__ import annotations

import os

import torch
import torch.distributed as dist


def tiles_of(width: int, height: int, tile: int) -> int:
    return ((width + tile - 1) // tile) * ((height + tile - 1) // tile)


def local_tile_count(total: int, rank: int, world: int) -> int:
    """Tiles rank owns under the round-robin deal (== rtgs_num_local_tiles)."""
    return 0 if rank >= total else (total - rank + world - 1) // world


def cost_sorted_order(width: int, height: int, tile: int, first: int, step: int, mode: str = "wedge"):
    """Launch order for a shard's local tiles (NumPy int32 permutation).  Scheduling heuristic
    only: the pixels do not depend on it.

    ``mode="centre"``: most central first.  On MI355X workgroups b, b+256, b+512, ... of one launch
    share a CU, so sorting a shard's tiles by expected cost hands every CU one tile of each cost
    class; without a measured cost, distance from the image centre is the prior (perspective piles
    depth complexity up there; the benchmark scenes are 5x heavier in the central half).

    ``mode="wedge"`` (default): workgroup b runs on XCD b % 8 (round-robin dispatch), and every XCD
    has its own 4 MiB L2.  The image is cut into 8 equal-count wedges around its centre, XCD x gets
    wedge x (position 8 i + x of the order holds the i-th most central tile of wedge x), so that an
    XCD's L2 serves the Gaussians (records, colour, SH) of one eighth of the screen instead of all
    of it, while every XCD still gets the same mix of heavy central and light border tiles.
    Measured on MI355X (k_trace): c3 0.140 -> 0.137 ms, c3 + SH 0.310 -> 0.289, c5 1.318 -> 1.301.
    Vertical strips or 4 x 2 blocks per XCD lose 20 % (the central regions are heavier); Morton
    order or radial bands inside a wedge lose 3-18 % against most-central-first.
    Computed by the library (rtgs_launch_order) so that C callers get the same order."""
    import numpy as np
    from . import _native as nat
    assert tile == nat.tile_size()
    n = local_tile_count(tiles_of(width, height, tile), first, step) if step > 0 else 0
    out = np.empty(max(n, 0), dtype=np.int32)
    if n > 0:
        got = nat.lib().rtgs_launch_order(width, height, first, step, 1 if mode == "wedge" else 0,
                                          out.ctypes.data)
        nat.check(min(got, 0), "rtgs_launch_order")
        assert got == n
    return out


class Partition:
    """Which rank renders which macro-tiles: ``tiles[r]`` = rank r's tile ids in compact-slot order, together a
    partition of 0..total-1.  ``round_robin`` is the arithmetic deal (tile t -> rank t mod world: the kernel's
    (tile_first, tile_step) form, equal shards, one ``dist.gather``); ``weighted`` gives every rank its own share
    (``RtgsTraceParams.tile_list``, point-to-point exchange of exactly the bytes each shard has).

    Why unequal shares: the destination receives world - 1 shards per frame, each over its own xGMI link, so a
    frame cannot land faster than ONE shard crosses one link -- 12.4 MB of a 1080p frame at world = 2, as long
    as a single MI355X needs to render the whole frame.  A destination that keeps the share x of the tiles that
    balances its own trace against one peer's transfer (``calibrated_shares``) shortens both."""

    MAX_MAPPED_RANKS = 16   # rtgs_frames_land_map / rtgs_untile_map pass the ranks' buffers as kernel arguments
    PERIOD = 64          # weighted deal: tile t belongs to the owner of residue t mod PERIOD
    _SPREAD = 37         # residues are dealt out in the order (i * 37) mod 64, so a rank's tiles do not form runs

    def __init__(self, tiles, total: int, name: str):
        import numpy as np
        self.tiles = [np.ascontiguousarray(t, dtype=np.int32) for t in tiles]
        self.world, self.total, self.name = len(self.tiles), int(total), name
        self.counts = [int(t.size) for t in self.tiles]
        seen = np.concatenate(self.tiles) if self.tiles else np.zeros(0, np.int32)
        if seen.size != self.total or not np.array_equal(np.sort(seen), np.arange(self.total, dtype=np.int32)):
            raise ValueError("not a partition of the frame's tiles")
        if self.world > 128 or self.total >= 1 << 24:
            raise ValueError("tile_map packs rank << 24 | slot")
        self.tile_map = np.empty(self.total, dtype=np.int32)
        for r, t in enumerate(self.tiles):
            self.tile_map[t] = (r << 24) | np.arange(t.size, dtype=np.int32)
        self.arithmetic = name == "round-robin"
        self.compact = False         # every rank's tiles form one compact part of the frame (wedges): region cull pays

    @classmethod
    def round_robin(cls, total: int, world: int) -> "Partition":
        import numpy as np
        return cls([np.arange(r, total, world, dtype=np.int32) for r in range(world)], total, "round-robin")

    @classmethod
    def weighted(cls, total: int, shares) -> "Partition":
        """Rank r gets about shares[r] (non-negative, any scale) of the tiles: whole residues mod PERIOD by largest
        remainder, dealt out in a spread order.  A rank may end up with no tiles at all."""
        import numpy as np
        sh = np.asarray(shares, dtype=np.float64)
        if sh.ndim != 1 or sh.size < 1 or not np.all(np.isfinite(sh)) or np.any(sh < 0) or sh.sum() <= 0:
            raise ValueError("shares must be non-negative and not all zero")
        want = sh / sh.sum() * cls.PERIOD
        m = np.floor(want).astype(np.int64)
        for r in np.argsort(-(want - m), kind="stable")[:cls.PERIOD - int(m.sum())]:
            m[r] += 1
        order = (np.arange(cls.PERIOD) * cls._SPREAD) % cls.PERIOD
        owner = np.empty(cls.PERIOD, dtype=np.int64)
        owner[order] = np.repeat(np.arange(sh.size), m)
        own = owner[np.arange(total) % cls.PERIOD]
        name = "weighted " + "/".join(str(int(v)) for v in m) + f" of {cls.PERIOD}"
        return cls([np.nonzero(own == r)[0].astype(np.int32) for r in range(sh.size)], total, name)

    @classmethod
    def wedges(cls, width: int, height: int, tile: int, world: int) -> "Partition":
        """One compact region per rank: equal-count angular sectors around the image centre (the construction
        rtgs_launch_order uses per XCD), so that every rank gets the same mix of heavy central and light border
        tiles of a centre-heavy scene -- and touches only the Gaussians of its sector.  Measured against the
        interleaved deal in tools/rank_loops.py (DEALS=wedges); not the default: its balance depends on the content."""
        import numpy as np
        tx, ty = (width + tile - 1) // tile, (height + tile - 1) // tile
        t = np.arange(tx * ty)
        ang = np.arctan2(((t // tx) + 0.5) * tile - 0.5 * height, ((t % tx) + 0.5) * tile - 0.5 * width)
        order = np.argsort(ang, kind="stable")
        owner = np.empty(t.size, dtype=np.int64)
        owner[order] = (np.arange(t.size) * world) // t.size
        part = cls([np.nonzero(owner == r)[0].astype(np.int32) for r in range(world)], t.size, f"{world} wedges")
        part.compact = True
        return part

    def shares(self):
        return [c / self.total for c in self.counts]


def balanced_share(world: int, t_shard: float, t_frame: float, t_xfer_shard: float) -> float:
    """The share x of the frame the destination should keep, from three times measured on it (any unit): tracing
    its 1/world shard, tracing the whole frame, and receiving one exchange of 1/world shards from its peers.
    Trace time is taken linear in the share, p + c x; an exchange of per-peer share y costs l y (the peers' links
    run in parallel); the destination's trace and the transfer overlap, so the frame rate is their maximum:
    p + c x = l (1 - x) / (world - 1).  Clamped to [1 / world, 1]: a link faster than the trace means an equal deal."""
    n = world
    if n <= 1:
        return 1.0
    c = max((t_frame - t_shard) / (1.0 - 1.0 / n), 0.0)
    p = max(t_shard - c / n, 0.0)
    l = max(t_xfer_shard, 0.0) * n
    lk = l / (n - 1)
    x = (lk - p) / (c + lk) if (c + lk) > 0 else 1.0 / n
    return min(max(x, 1.0 / n), 1.0)


def _phase(*what):
    """RTGS_TRACE_PHASES=1: one line on stderr per phase of the multi-rank set-up (a fault or a hang then names its phase)."""
    if os.environ.get("RTGS_TRACE_PHASES") == "1":
        import sys
        print(f"[rtgs rank {os.environ.get('RANK', '?')}]", *what, file=sys.stderr, flush=True)


class _Done:
    def wait(self):
        return True


class _StagedRecv:
    """A receive into pinned host memory that lands in its device buffer when it is waited for."""

    def __init__(self, work, host, device):
        self.work, self.host, self.device = work, host, device

    def wait(self):
        self.work.wait()
        self.device.copy_(self.host, non_blocking=True)     # (ordered on the current stream, before the scatter)
        return True


def _no_gather(tensor, gather_list=None, dst=0, group=None, async_op=False):
    """Stand-in for dist.gather while the compute side of the loop is being timed."""
    return _Done()


def calibrated_shares(backend, rank: int, world: int, dst: int = 0, group=None, frames: int = 24) -> dict:
    """Shares for ``Partition.weighted`` from three times measured here, now, with the real loop (no link rate is
    assumed): the destination's loop over its 1/world shard with the exchange stubbed out (project + trace with the
    frames in flight + untile), its loop over whole frames, and one exchange of 1/world shards.  Collective: every
    rank calls it at the same time; the destination's figures are the ones used and broadcast."""
    import time
    sync = getattr(backend, "synchronize", lambda: None)

    def loop(fn, n):
        fn()
        sync()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        sync()
        return (time.perf_counter() - t0) / n

    def frames_of(r, n):
        def run():
            for _ in range(4):
                r.render_frame()
            r.finish()
        t = loop(run, max(n // 4, 1)) / 4
        r.close()
        return t

    _phase("calibration: shard loop")
    t_shard = frames_of(TileParallelRenderer(backend=backend, rank=rank, world=world, dst=dst, group=group,
                                             gather_fn=_no_gather, partition="equal"), frames)
    _phase("calibration: frame loop")
    t_frame = frames_of(TileParallelRenderer(backend=backend, rank=0, world=1), max(frames // 2, 4))
    _phase("calibration: exchange")
    probe = TileParallelRenderer(backend=backend, rank=rank, world=world, dst=dst, group=group, partition="equal",
                                 frames_in_flight=1, gather_batch=1)
    buf, glist = probe.compact[0], (probe._glist[0] if probe._glist else None)
    dist.barrier(group)
    t_x = loop(lambda: dist.gather(buf, gather_list=glist, dst=dst, group=group), frames)
    probe.close()
    _phase("calibration: broadcast", t_shard, t_frame, t_x)
    vals = [[t_shard, t_frame, t_x]]
    dist.broadcast_object_list(vals, src=dst, group=group)
    t_shard, t_frame, t_x = vals[0]
    x = balanced_share(world, t_shard, t_frame, t_x)
    shares = [(1.0 - x) / (world - 1)] * world
    shares[dst] = x
    return dict(x=x, shares=shares, equal=bool(x <= 1.02 / world), t_shard_ms=t_shard * 1e3, t_frame_ms=t_frame * 1e3,
                t_exchange_ms=t_x * 1e3)


class HipBackend:
    """The product backend: HIP kernels through the C ABI (include/rtgs_hip.h)."""

    def __init__(self, scene, camera, depth: int, t_cutoff: float = 0.0):
        from . import _native as nat
        self.nat = nat
        self.scene, self.camera = scene, camera
        self.depth, self.t_cutoff = int(depth), float(t_cutoff)
        # RTGS_TILE_ORDER: 0 = launch order is tile order, centre | wedge (default) see cost_sorted_order
        self.tile_order_mode = os.environ.get("RTGS_TILE_ORDER", "wedge")
        self.use_tile_order = self.tile_order_mode != "0"
        self.device = nat.device()
        self.width, self.height = camera.buf_size.x, camera.buf_size.y
        self.tile = nat.tile_size()
        need = nat.lib().rtgs_workspace_bytes_frame(scene.num_gaussians, self.width, self.height)
        self.workspace = torch.empty(need, dtype=torch.uint8, device=self.device)
        self._orders = {}
        self._state = {}
        self._scene_native = self._camera_native = None
        # (round 5: the hipGraph replay of project + trace and the projection on a stream of its own -- RTGS_GRAPH /
        # RTGS_PIPELINE, measured slower in rounds 2, 3 and 4, HISTORY.md -- are gone)

    def scene_native(self):
        """``scene.native()`` cached per commit (the struct only holds the streamed tensors' addresses)."""
        sc = self.scene
        if self._scene_native is None or self._scene_native[0] != sc.version:
            self._scene_native = (sc.version, sc.native())
        return self._scene_native[1]

    def camera_native(self):
        """``camera.native()`` cached per pose: the frame loop rebuilds it only when the camera moved."""
        c = self.camera
        key = (bytes(memoryview(c.position)), bytes(memoryview(c.rotation)), bytes(memoryview(c.focal_length)))
        if self._camera_native is None or self._camera_native[0] != key:
            self._camera_native = (key, c.native())
        return self._camera_native[1]

    def tile_order(self, tile_first: int, tile_step: int):
        key = (tile_first, tile_step)
        if key not in self._orders:
            o = cost_sorted_order(self.width, self.height, self.tile, tile_first, tile_step,
                                  "centre" if self.tile_order_mode == "centre" else "wedge")
            self._orders[key] = torch.from_numpy(o).to(self.device)
        return self._orders[key]

    def shard(self, tiles, region: bool = False):
        """Device copies of an explicit tile list and of its launch order: what ``trace(shard=...)`` takes.
        ``region=True``: also the rectangle of pixel-centre offsets that contains the tiles
        (``RtgsTraceParams.region``: the shard's projection then skips the Gaussians that cannot reach it)."""
        import numpy as np
        nat = self.nat
        tiles = np.ascontiguousarray(tiles, dtype=np.int32)
        rect = None
        if region and tiles.size:
            T, tx = self.tile, (self.width + self.tile - 1) // self.tile
            i0, j0 = (tiles % tx) * T, (tiles // tx) * T
            rect = (float(i0.min()) + 0.5 - 0.5 * self.width, float(j0.min()) + 0.5 - 0.5 * self.height,
                    float(np.minimum(i0 + T, self.width).max()) - 0.5 - 0.5 * self.width,
                    float(np.minimum(j0 + T, self.height).max()) - 0.5 - 0.5 * self.height)
        order = None
        if self.use_tile_order and tiles.size:
            o = np.empty(tiles.size, dtype=np.int32)
            got = nat.lib().rtgs_launch_order_list(self.width, self.height, tiles.ctypes.data, tiles.size,
                                                   0 if self.tile_order_mode == "centre" else 1, o.ctypes.data)
            nat.check(min(got, 0), "rtgs_launch_order_list")
            order = torch.from_numpy(o).to(self.device)
        return dict(list=torch.from_numpy(tiles).to(self.device), count=int(tiles.size), order=order, region=rect)

    def shard_params(self, shard, flags: int):
        """RtgsTraceParams of an explicit shard (``shard()``)."""
        import ctypes
        nat = self.nat
        flags |= nat.TRACE_TILE_LIST
        rect = shard.get("region")
        if rect is not None:
            flags |= nat.TRACE_REGION
        return nat.RtgsTraceParams(self.depth, self.t_cutoff, 0, 1, flags, nat.ptr(shard["order"]),
                                   nat.ptr(shard["list"]) if shard["count"] else None, shard["count"],
                                   (ctypes.c_float * 4)(*(rect or (0.0, 0.0, 0.0, 0.0))))

    def alloc(self, *shape):
        return torch.zeros(shape, dtype=torch.float32, device=self.device)

    def to_device(self, array):
        return torch.from_numpy(array).to(self.device)

    def synchronize(self):
        torch.cuda.synchronize(self.device)

    def project(self):
        nat = self.nat
        nat.check(nat.lib().rtgs_project(self.scene.native(), self.camera.native(),
                                         nat.ptr(self.workspace), self.workspace.numel(),
                                         nat.stream()), "rtgs_project")

    def trace(self, out, tile_first: int, tile_step: int, compact: bool, projected: bool = False,
              workspace=None, stream=None, shard=None):
        """stream: a torch.cuda.Stream to launch on instead of the current one (the C ABI takes the stream as
        an argument: no need to make it current)."""
        nat = self.nat
        ws = self.workspace if workspace is None else workspace
        st = nat.stream() if stream is None else stream.cuda_stream
        flags = nat.TRACE_COMPACT if compact else 0
        if shard is not None:      # an explicit tile list (``shard()``) instead of the (first, step) deal
            params = self.shard_params(shard, flags)
        else:
            order = self.tile_order(tile_first, tile_step) if self.use_tile_order else None
            params = nat.RtgsTraceParams(self.depth, self.t_cutoff, tile_first, tile_step, flags,
                                         nat.ptr(order))
        att = start = None
        if self.depth > nat.MAX_DEPTH:
            # a chain of launches through per-pixel state (rtgs_hip.h, RtgsTraceParams.depth): one
            # (attenuation, ray.start) pair per workspace, i.e. per frame in flight
            pair = self._state.get(ws.data_ptr())
            if pair is None:
                pair = self._state[ws.data_ptr()] = (self.alloc(self.width, self.height),
                                                     self.alloc(self.width, self.height))
                if stream is not None:   # the zero-fills ran on the current stream: order the first use after them
                    stream.wait_stream(nat.current_stream(self.device.index))
            att, start = pair
        nat.check(nat.lib().rtgs_trace(self.scene.native(), self.camera.native(), params,
                                       nat.ptr(out), nat.ptr(att), nat.ptr(start), 1, nat.ptr(ws), ws.numel(),
                                       1 if projected else 0, st),
                  "rtgs_trace")

    def render(self, out, tile_first: int, tile_step: int, compact: bool, stream=None, workspace=None, shard=None):
        """project + trace of one shard into `out`: two (three with bins) direct launches on `stream` (default: the
        current stream) with `workspace` (default: the backend's own)."""
        self.trace(out, tile_first, tile_step, compact, stream=stream, workspace=workspace, shard=shard)

    # ---- B frames per library call ------------------------------------------------------------
    def camera_array(self, poses=None, count: int = 1):
        """A HOST array of RtgsCamera for ``rtgs_shard_frames``: one per pose of `poses` -- a sequence of
        (position, rotation) or an (n, 7) array x y z qx qy qz qw -- or `count` copies of the camera's current pose.
        Filled through NumPy (one structured assignment, not n x 11 ctypes stores); returns (array, n): keep the
        array alive until the call has returned."""
        import numpy as np
        nat = self.nat
        c = self.camera
        if poses is None:
            key = (count, bytes(memoryview(c.position)), bytes(memoryview(c.rotation)), bytes(memoryview(c.focal_length)))
            hit = getattr(self, "_cam_array", None)
            if hit is not None and hit[0] == key:
                return hit[1], count
            pose = np.concatenate([np.asarray(c.position, np.float32), np.asarray(c.rotation, np.float32)])[None, :]
            pose = np.repeat(pose, count, axis=0)
        else:
            pose = _pose_rows(poses)
            key = None
        n = int(pose.shape[0])
        arr = np.zeros(n, dtype=nat.CAMERA_DTYPE)
        arr["position"], arr["rotation"] = pose[:, :3], pose[:, 3:]
        arr["width"], arr["height"] = self.width, self.height
        arr["fx"], arr["fy"] = float(c.focal_length[0]), float(c.focal_length[1])
        if key is not None:
            self._cam_array = (key, arr)
        return arr, n

    def untile(self, gathered, world: int, tiles_per_rank: int, image, stream=None):
        nat = self.nat
        st = nat.stream() if stream is None else stream.cuda_stream
        nat.check(nat.lib().rtgs_untile(nat.ptr(gathered), world, tiles_per_rank, self.width,
                                        self.height, nat.ptr(image), st), "rtgs_untile")

    def untile_map(self, bufs, counts, tile_map, image, stream=None):
        """The landing side of any ``Partition``: bufs[r] = rank r's compact buffer (None for an empty shard)."""
        import ctypes
        nat = self.nat
        st = nat.stream() if stream is None else stream.cuda_stream
        ptrs = (ctypes.c_void_p * len(bufs))(*[None if b is None else b.data_ptr() for b in bufs])
        cnt = (ctypes.c_int32 * len(counts))(*counts)
        nat.check(nat.lib().rtgs_untile_map(ptrs, cnt, len(bufs), nat.ptr(tile_map), self.width, self.height,
                                            nat.ptr(image), st), "rtgs_untile_map")


_STREAM_POOL = {}


def _render_streams(device, count: int, offset: int):
    """`count` render streams out of ONE pool per (process, device), starting at `offset`: renderers built one after the
    other in a process (bench.py times five deals) share their streams instead of drawing new ones each -- which HIP
    stream shares a hardware queue with which other is a matter of creation order, and a render stream that shares the
    caller's queue serialises behind the exchange's event waits (round 5: the same rank's loop alternated between 25 and
    28-35 us per c3 frame from one renderer to the next; ``TileParallelRenderer`` times the candidates once and keeps
    the fastest set, ``_pick_render_streams``)."""
    pool = _STREAM_POOL.setdefault((device.type, device.index), [])
    while len(pool) < offset + count:
        pool.append(torch.cuda.Stream(device=device))
    return pool[offset:offset + count]


def _pose_rows(poses):
    """(n, 7) float32 rows x y z qx qy qz qw from (position, rotation) pairs or an array of that shape."""
    import numpy as np
    if isinstance(poses, (list, tuple)) and poses and len(poses[0]) == 2:
        poses = [[*p[0], *p[1]] for p in poses]
    rows = np.ascontiguousarray(poses, dtype=np.float32).reshape(-1, 7) if len(poses) else np.zeros((0, 7), np.float32)
    return rows


def _no_all_to_all(output, input, output_split_sizes=None, input_split_sizes=None, group=None):
    """Stand-in for dist.all_to_all_single while the compute side of the loop is being timed."""
    return None


def _all_to_all(output, input, output_split_sizes=None, input_split_sizes=None, group=None):
    """``dist.all_to_all_single`` on the current stream (RCCL: one grouped send / recv to and from every peer, all links
    at once).  Over gloo with device tensors -- the one-GPU rehearsal of the N > 1 path, never a measurement --
    the blocks are staged through the host."""
    if input.is_cuda and dist.get_backend(group) == "gloo":
        host_out = torch.empty(output.shape, dtype=output.dtype)
        dist.all_to_all_single(host_out, input.cpu(), output_split_sizes, input_split_sizes, group=group)
        output.copy_(host_out)
        return None
    return dist.all_to_all_single(output, input, output_split_sizes, input_split_sizes, group=group)


class FrameBatchRenderer:
    """Whole frames on ONE GPU, up to `frames` consecutive poses per LAUNCH (grid.y = frames through
    ``rtgs_shard_frames``): a turntable or a benchmark that knows its next poses pays one k_project / k_bin / k_trace
    launch -- and their latency-bound heads and tails -- per batch instead of per frame (c3: the three launches of a
    frame are 0.147 ms one by one, of which k_project + k_bin are 17 us of mostly latency).  ``streams=2``: consecutive
    calls alternate over two render streams, each with a workspace and a set of images of its own, so that batch k + 1's
    projection and bins run under batch k's trace -- pays for short frames (c2: 0.0295 -> 0.0264 ms per frame) and not
    at c3 (0.1318 -> 0.1343: what a dependency across two hardware queues costs, again).  Every
    frame is fully projected, binned and traced; ``images[k]`` is frame k of the LAST ``render_frames`` call, complete
    in the caller's stream order, and stays valid until the call after the next.  The reference's viewer renders one
    pose per loop iteration (__main__.py:236-257): this is for consumers that take frames in batches, at frames - 1
    frames of latency."""

    def __init__(self, scene, camera, depth: int = 16, frames: int = 8, t_cutoff: float = 0.0, streams: int = 1):
        import ctypes
        self.backend = b = HipBackend(scene, camera, depth, t_cutoff)
        nat = b.nat
        lib = nat.lib()
        self.frames = max(1, min(int(frames), lib.rtgs_max_frames_per_launch()))
        self._n = S = max(1, int(streams))
        self._images = [b.alloc(self.frames, b.width, b.height, 3) for _ in range(S)]
        need = lib.rtgs_workspace_bytes_frames(scene.num_gaussians, b.width, b.height, self.frames)
        self._ws = [torch.empty(need, dtype=torch.uint8, device=b.device) for _ in range(S)]
        order = b.tile_order(0, 1) if b.use_tile_order else None
        self._params = nat.RtgsTraceParams(b.depth, b.t_cutoff, 0, 1, 0, nat.ptr(order))
        self._order = order

        def event():
            h = ctypes.c_void_p()
            nat.check(lib.rtgs_event_create(ctypes.byref(h)), "rtgs_event_create")
            return h.value

        self._done = [event() for _ in range(S)]
        # free[k]: "the caller's stream is past everything it queued on set k's images" -- recorded at the START of the
        # call after the one that rendered set k, i.e. behind that batch's consumers but AHEAD of the wait for the next
        # batch's trace, so that the batch after next can start projecting while that trace still runs
        self._free = [torch.cuda.Event() for _ in range(S)]
        self._free_set = [False] * S
        self._streams = _render_streams(b.device, S, 0) if S > 1 else None
        self._done_c = [(ctypes.c_void_p * 1)(e) for e in self._done]
        self._ws_c = [(ctypes.c_void_p * 1)(w.data_ptr()) for w in self._ws]
        self._outs_c = []
        for img in self._images:
            stride = img.stride(0) * 4
            self._outs_c.append((ctypes.c_void_p * self.frames)(*[img.data_ptr() + k * stride for k in range(self.frames)]))
        self._lib, self._turn, self._last, self._seen = lib, 0, 0, [None] * S

    @property
    def images(self):
        """(frames, W, H, 3): the images of the last ``render_frames`` call."""
        return self._images[self._last]

    def render_frames(self, poses=None, count: int | None = None) -> int:
        """Render up to `frames` poses ((position, rotation) pairs or (n, 7) rows; default: `count` copies of the
        camera's pose) into ``images[0 .. n)``; returns n."""
        import ctypes
        b, nat = self.backend, self.backend.nat
        rows = _pose_rows(poses) if poses is not None else None
        n = len(rows) if rows is not None else int(self.frames if count is None else count)
        if n < 1 or n > self.frames:
            raise ValueError(f"1 .. {self.frames} frames per call")
        cams, _ = b.camera_array(rows, n)
        k = self._turn % self._n
        self._turn += 1
        cur = nat.stream()
        if self._streams is None:
            st = cur
        else:
            rs = self._streams[k]
            st = rs.cuda_stream
            key = b.scene.content_key()
            if self._seen[k] != key:       # scene uploads / edits queued on the caller's stream
                rs.wait_stream(nat.current_stream(b.device.index))
                self._seen[k] = key
            if self._turn > 1:
                prev = (k - 1) % self._n
                self._free[prev].record(nat.current_stream(b.device.index))
                self._free_set[prev] = True
        streams = (ctypes.c_void_p * 1)(st)
        nat.check(self._lib.rtgs_shard_frames(b.scene_native(), cams.ctypes.data, n, self._params, self._outs_c[k], None, None,
                                              self._ws_c[k], self._ws[k].numel(), streams, 1, 0,
                                              self._free[k].cuda_event if (self._streams is not None and self._free_set[k]) else None,
                                              self._done_c[k],
                                              cur, 1 if self._streams is not None else 0), "rtgs_shard_frames")
        self._last = k
        return n

    def close(self) -> None:
        evs, self._done = self._done, []
        if evs:
            torch.cuda.synchronize(self.backend.device)
            for e in evs:
                self._lib.rtgs_event_destroy(e)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TileParallelRenderer:
    """Renders whole frames on `world` ranks; where they land is ``landing``:

    ``"root"`` (default): every frame in ``framebuffer`` on rank `dst`, by one gather per batch.
    ``"rotate"``: the roots rotate -- of every batch of B = k * world frames the k frames [d k, (d + 1) k) land on
    rank d (frame f on rank f mod world when B = world), by ONE ``all_to_all_single`` per batch: every rank is the
    root of k of the batch's frames (``landed_frames()`` names them, ``destination(f)`` says where frame f goes),
    every xGMI link carries 1/world of the single root's inbound traffic and every rank untiles one frame in world.

    ``frames_in_flight`` (default 2, ``RTGS_FRAMES_IN_FLIGHT``): a 1/8 shard of a 1080p frame is ONE
    resident round of workgroups, so its trace takes as long as its heaviest tile's dependency chain
    (37 us at c3 whether the shard is 1/8 or 1/16 of the frame) and most of the chip idles behind it.
    With F frames in flight every rank alternates its frames over F render streams, each with its
    own workspace, so consecutive frames' traces overlap (two 1/8-shard traces: 43.5 us per pair
    against 74.3 one after the other, tools/concurrency_probe.py) and the loop is bound by
    throughput, not by one tile's latency.  The exchange and the untile stay on the caller's stream, in frame order.

    ``gather_batch`` (``RTGS_GATHER_BATCH``; default 1 for "root", the smallest multiple of world >= 8 for
    "rotate", where it is rounded up to a multiple of world): B consecutive frames' shards travel in one
    collective and land in a ring of framebuffers, the newest being ``framebuffer``: the collective's fixed host
    and launch cost (16-26 us per call, more than a 1/8-shard trace of a 1080p frame) is paid once per B frames,
    at the price of B - 1 frames of latency.  ``finish()`` flushes a partial batch.

    ``render_frames(poses)`` queues many frames at once: per batch ONE library call on the render side
    (``rtgs_shard_frames``: every frame's project + trace, spread over the F render streams, from a host array
    of poses), one collective, one landing call (``rtgs_frames_land``).  ``render_frame()`` is
    ``render_frames(count=1)`` with the camera's current pose."""

    def __init__(self, scene=None, camera=None, depth: int = 16, rank: int = 0, world: int = 1,
                 t_cutoff: float = 0.0, dst: int = 0, group=None, backend=None, gather_fn=None,
                 frames_in_flight: int | None = None, gather_batch: int | None = None, partition=None, p2p_fn=None,
                 landing: str = "root", a2a_fn=None):
        """``partition``: None / "equal" = the round-robin deal (equal shards, one collective); a ``Partition``
        or a sequence of per-rank shares = that deal (exchange of each shard's own bytes); "auto" =
        ``calibrated_shares`` (landing "root" only -- with rotating roots every rank is a destination as often as
        any other and the equal deal is the balanced one; every rank must construct its renderer at the same time:
        it times one shard trace, one frame trace and one exchange on this machine and the destination broadcasts
        the shares)."""
        self.backend = backend if backend is not None else HipBackend(scene, camera, depth, t_cutoff)
        self._gather = gather_fn if gather_fn is not None else dist.gather   # injectable for tests
        self._p2p_fn = p2p_fn                                                # likewise: (mine, landing) -> works
        self._a2a = a2a_fn if a2a_fn is not None else _all_to_all            # likewise: dist.all_to_all_single's signature
        self.rank, self.world, self.dst, self.group = int(rank), int(world), int(dst), group
        if landing not in ("root", "rotate"):
            raise ValueError('landing: "root" or "rotate"')
        self.landing = landing if self.world > 1 else "root"
        rot = self.landing == "rotate"
        b = self.backend
        self.TILE = int(b.tile)
        self.total_tiles = tiles_of(b.width, b.height, self.TILE)
        self.calibration = None
        if isinstance(partition, str) and partition == "wedges":
            partition = Partition.wedges(b.width, b.height, self.TILE, self.world)
        if isinstance(partition, str) and partition == "auto":
            partition = None
            if self.world > 1 and not rot:
                self.calibration = calibrated_shares(b, self.rank, self.world, self.dst, self.group)
                if not self.calibration["equal"]:
                    partition = self.calibration["shares"]
        if partition is None or (isinstance(partition, str) and partition == "equal"):
            partition = Partition.round_robin(self.total_tiles, self.world)
        elif not isinstance(partition, Partition):
            partition = Partition.weighted(self.total_tiles, partition)
        if partition.world != self.world or partition.total != self.total_tiles:
            raise ValueError("partition is of another frame / world size")
        if not partition.arithmetic and self.world > Partition.MAX_MAPPED_RANKS:
            raise ValueError(f"an explicit tile deal lands through rtgs_frames_land_map, which takes at most "
                             f"{Partition.MAX_MAPPED_RANKS} ranks; world = {self.world} (the round-robin deal has no limit)")
        _phase("deal", partition.name, "landing", self.landing, "local tiles", partition.counts[self.rank])
        self.partition = partition
        self.tiles_per_rank = (self.total_tiles + self.world - 1) // self.world     # the arithmetic deal's padded shard
        self.local_tiles = partition.counts[self.rank]
        is_dst = self.world == 1 or rot or self.rank == self.dst
        # any other deal: this rank's tile list (device), the frame's tile -> (rank, slot) map (destinations)
        self._shard = self._tile_map = None
        if not partition.arithmetic:
            self._shard = (b.shard(partition.tiles[self.rank], region=partition.compact) if hasattr(b, "shard")
                           else dict(tiles=partition.tiles[self.rank]))
            if is_dst:
                self._tile_map = b.to_device(partition.tile_map) if hasattr(b, "to_device") else partition.tile_map
        if frames_in_flight is None:
            frames_in_flight = int(os.environ.get("RTGS_FRAMES_IN_FLIGHT", "2"))
        if gather_batch is None:
            env = os.environ.get("RTGS_GATHER_BATCH")
            gather_batch = int(env) if env else (max(8, self.world) if rot else 1)
        if frames_in_flight < 1 or gather_batch < 1:
            raise ValueError("frames_in_flight and gather_batch must be >= 1")
        streams = (self.world > 1 and isinstance(b, HipBackend) and
                   os.environ.get("RTGS_RENDER_STREAM", "1") != "0")
        self.frames_in_flight = int(frames_in_flight) if self.world > 1 else 1
        N = self.world
        if rot:
            gather_batch = (int(gather_batch) + N - 1) // N * N      # k whole frames per destination and batch
        self.gather_batch = int(gather_batch) if (streams or rot) else 1
        self.per_dest = self.gather_batch // N if rot else self.gather_batch    # k: frames a destination lands per batch
        self._fbs = [b.alloc(b.width, b.height, 3) for _ in range(self.per_dest)] if is_dst else []
        self._landed = 0            # which of _fbs holds the newest frame
        self._landed_frames = []    # (frame number, index into _fbs) of the batch that landed last on this rank
        self._pending = []          # one-stream form: (work, slot) of the gathers not yet completed
        self._frame = 0
        self._comm = None
        if self.world > 1:
            F, B, T = self.frames_in_flight, self.gather_batch, self.TILE
            arith = partition.arithmetic
            E = T * T * 3
            # rotating roots exchange exact sizes (all_to_all_single with split sizes); the single root's point-to-point
            # buffers keep at least one tile so that no tensor is empty
            own = self.tiles_per_rank if arith else (self.local_tiles if rot else max(self.local_tiles, 1))
            shard = (own, T, T, 3)
            peer = lambda r: (max(partition.counts[r], 1), T, T, 3)
            self._split_out = self._split_in = self._recv_off = None
            if rot:
                k = self.per_dest
                self._slots = ((2 * F + B - 1) // B + 1) if streams else 2
                self.compact = [b.alloc(B, *shard) for _ in range(self._slots)]
                if arith:
                    self.gathered = [b.alloc(N * k, *shard) for _ in range(2)]           # [source][k][tiles][T][T][3]
                else:
                    self._split_out = [k * c * E for c in partition.counts]
                    self._split_in = [k * self.local_tiles * E] * N
                    self._recv_off = [sum(self._split_out[:r]) for r in range(N)]
                    self.gathered = [b.alloc(max(sum(self._split_out), 1)) for _ in range(2)]   # flat, by source
                self._glist = []
            elif streams:
                # batch-slot ring: deep enough that a render stream never waits for the gather of the
                # batch it traced into last (only for one S batches back, long done)
                self._slots = (2 * F + B - 1) // B + 1
                self.compact = [b.alloc(B, *shard) for _ in range(self._slots)]
                # gather and landing of consecutive batches are serial on the caller's stream: two
                # landing buffers are plenty
                if arith:
                    self.gathered = [b.alloc(self.world, B, *shard) for _ in range(2)] if is_dst else []
                else:       # one landing buffer per peer, sized for that peer's share
                    self.landing_bufs = [[None if r == self.rank else b.alloc(B, *peer(r)) for r in range(self.world)]
                                         for _ in range(2)] if is_dst else []
                    self.gathered = []
            else:
                # one-stream form: at most F gathers outstanding, one slot and landing buffer each
                self._slots = max(F, 2)
                self.compact = [b.alloc(*shard) for _ in range(self._slots)]
                if arith:
                    self.gathered = [b.alloc(self.world, *shard) for _ in range(self._slots)] if is_dst else []
                else:
                    self.landing_bufs = [[None if r == self.rank else b.alloc(*peer(r)) for r in range(self.world)]
                                         for _ in range(self._slots)] if is_dst else []
                    self.gathered = []
            if not rot:
                # the per-rank views torch.distributed.gather wants, made once (8 views cost ~8 us per frame)
                self._glist = [list(g.unbind(0)) for g in self.gathered]
            if streams:
                self._comm = self._make_stream_state(F, B)
                self._pick_render_streams()

    @property
    def framebuffer(self):
        """(W, H, 3) image of the newest frame that has landed on this rank (None on a rank nothing lands on)."""
        return self._fbs[self._landed] if self._fbs else None

    def landed_frames(self):
        """[(frame number, (W, H, 3) image)] of the batch that landed last on this rank, oldest first; frame
        numbers count ``render_frame`` calls from 0.  Valid in the caller's stream order until the next batch lands."""
        return [(f, self._fbs[i]) for f, i in self._landed_frames]

    def destination(self, frame: int) -> int:
        """The rank frame number `frame` lands on.  Rotating roots: the k = B / world frames [d k, (d + 1) k) of every
        batch land on rank d -- frame f on rank f mod world when a batch is world frames -- so that a rank's block of
        the batch's send buffer is k CONSECUTIVE frames (all_to_all_single wants one contiguous block per destination,
        and consecutive frames of equal spacing can share a launch)."""
        if self.landing == "rotate":
            return (frame % self.gather_batch) // self.per_dest
        return self.dst

    def _batch_position(self, j: int) -> int:
        """Where frame j of a batch sits in the batch's send buffer (in batch order, for both landings)."""
        return j

    def _lands_here(self, n_valid: int) -> int:
        """How many of a batch's first n_valid frames land on this rank."""
        if self.landing == "rotate":
            return max(0, min(self.per_dest, n_valid - self.rank * self.per_dest))
        return n_valid if self._fbs else 0

    def _note_landed(self, batch: int, n_land: int) -> None:
        if n_land:
            B = self.gather_batch
            first = self.rank * self.per_dest if self.landing == "rotate" else 0
            self._landed_frames = [(batch * B + first + i, i) for i in range(n_land)]
            self._landed = n_land - 1

    def _make_stream_state(self, F: int, B: int):
        """F render streams (handed to the C ABI as arguments, never made current) with a workspace
        each; per batch slot and render stream a `done` event, per batch slot a `sent` event (raw hipEvent_t
        handles: the library records and waits on them inside rtgs_shard_frames / rtgs_frames_land)."""
        import ctypes
        b = self.backend
        nat = b.nat
        lib = nat.lib()

        def event():
            h = ctypes.c_void_p()
            nat.check(lib.rtgs_event_create(ctypes.byref(h)), "rtgs_event_create")
            return h.value

        # RTGS_SHARD_BINS=0 (experiments): this rank's shards scan the slab bounds themselves instead of reading the
        # per-frame bins -- exact ties may then resolve differently from a frame rendered with them
        flags = nat.TRACE_COMPACT | (nat.TRACE_BINS_OFF if os.environ.get("RTGS_SHARD_BINS", "1") == "0" else 0)
        if self._shard is not None:
            order = self._shard["order"]
            params = b.shard_params(self._shard, flags)
        else:
            order = b.tile_order(self.rank, self.world) if (b.use_tile_order and self.local_tiles) else None
            params = nat.RtgsTraceParams(b.depth, b.t_cutoff, self.rank, self.world, flags, nat.ptr(order))
        state = [(None, None)] * F
        if b.depth > nat.MAX_DEPTH:      # chained launches need per-pixel state, one pair per frame in flight
            state = [(b.alloc(b.width, b.height), b.alloc(b.width, b.height)) for _ in range(F)]
        # Frames per LAUNCH (RTGS_MERGE_FRAMES, default: as many of a batch as the library takes, 8): every render
        # stream's workspace holds that many frames' regions, and rtgs_shard_frames renders runs of consecutive frames
        # with one k_project / k_bin / k_trace each (grid.y = frames) -- a 1/8 shard of a 1080p frame is one resident
        # round of workgroups whose trace takes as long as its heaviest tile; eight of them in one grid fill the chip
        merge = int(os.environ.get("RTGS_MERGE_FRAMES", "0")) or lib.rtgs_max_frames_per_launch()
        merge = max(1, min(merge, B, lib.rtgs_max_frames_per_launch()))
        if b.depth > nat.MAX_DEPTH or b.depth < 2 or b.t_cutoff > 0:
            merge = 1                    # (chained launches, the single-peel and the cut-off forms render frame by frame)
        self.frames_per_launch = merge
        if merge > 1:
            need = lib.rtgs_workspace_bytes_frames(b.scene.num_gaussians, b.width, b.height, merge)
            ws = [torch.empty(need, dtype=torch.uint8, device=b.device) for _ in range(F)]
        else:
            ws = [b.workspace] + [torch.empty_like(b.workspace) for _ in range(F - 1)]
        images = (ctypes.c_void_p * len(self._fbs))(*[fb.data_ptr() for fb in self._fbs]) if self._fbs else None
        self._counts_c = (ctypes.c_int32 * self.world)(*self.partition.counts)
        render = _render_streams(b.device, F, 0)
        done = [[event() for _ in range(F)] for _ in range(self._slots)]
        return dict(render=render, ws=ws, state=state, done=done,
                    done_c=[(ctypes.c_void_p * F)(*row) for row in done],
                    streams_c=(ctypes.c_void_p * F)(*[rs.cuda_stream for rs in render]),
                    ws_c=(ctypes.c_void_p * F)(*[w.data_ptr() for w in ws]),
                    att_c=(ctypes.c_void_p * F)(*[nat.ptr(a) for a, _ in state]) if state[0][0] is not None else None,
                    start_c=(ctypes.c_void_p * F)(*[nat.ptr(st) for _, st in state]) if state[0][0] is not None else None,
                    sent=[event() for _ in range(self._slots)], seen=[None] * F, turn=0,
                    params=params, order=order, images=images, g=0, lib=lib)

    _PICKED = {}          # (device, F) -> offset into the stream pool of the set that timed fastest in this process

    def _pick_render_streams(self) -> None:
        """Which F streams of the process's pool render: the set under which THIS rank's loop -- render side and landing,
        the exchange left out -- runs fastest, timed once per (process, device, F) over three candidate sets (48 frames
        each, a few milliseconds).  HIP maps streams to hardware queues by creation order; a render stream that shares
        its queue with the caller's stream waits behind the exchange's barrier packets.  RTGS_STREAM_PICK=0: the first set."""
        import time
        import ctypes
        cm, b = self._comm, self.backend
        F = self.frames_in_flight
        key = (b.device.index, F)
        if os.environ.get("RTGS_STREAM_PICK", "1") == "0" or self.frames_per_launch < 2 or not self.local_tiles:
            return
        if key not in TileParallelRenderer._PICKED:
            saved = (self._a2a, self._gather, self._p2p_fn)
            self._a2a, self._gather, self._p2p_fn = _no_all_to_all, _no_gather, (lambda mine, landing: [])
            best = None
            try:
                for cand in range(3):
                    rs = _render_streams(b.device, F, cand * F)
                    cm["render"], cm["streams_c"] = rs, (ctypes.c_void_p * F)(*[x.cuda_stream for x in rs])
                    cm["seen"] = [None] * F
                    n = 6 * self.gather_batch
                    self.render_frames(count=self.gather_batch)
                    self.finish()
                    torch.cuda.synchronize(b.device)
                    t0 = time.perf_counter()
                    self.render_frames(count=n)
                    self.finish()
                    torch.cuda.synchronize(b.device)
                    dt = time.perf_counter() - t0
                    if best is None or dt < best[0]:
                        best = (dt, cand)
            finally:
                self._a2a, self._gather, self._p2p_fn = saved
            TileParallelRenderer._PICKED[key] = best[1] * F
            _phase("render streams: candidate set", best[1], "of 3")
            self._frame = 0
            self._landed_frames = []
            cm["turn"] = 0
        rs = _render_streams(b.device, F, TileParallelRenderer._PICKED[key])
        cm["render"], cm["streams_c"] = rs, (ctypes.c_void_p * F)(*[x.cuda_stream for x in rs])
        cm["seen"] = [None] * F

    def close(self) -> None:
        """Drain the render streams and release the events (also runs when the object is collected)."""
        cm, self._comm = self._comm, None
        if cm is not None:
            for rs in cm["render"]:
                rs.synchronize()
            for ev in [e for row in cm["done"] for e in row] + cm["sent"]:
                cm["lib"].rtgs_event_destroy(ev)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- frame loop ------------------------------------------------------------------------
    def _scene_key(self):
        """Changes when work queued on the CALLER's stream may have written what the render streams
        read: a re-commit, or an in-place edit of a streamed tensor (torch bumps ``_version``)."""
        return self.backend.scene.content_key()

    def render_frame(self) -> None:
        """Queue one frame of the camera's current pose (``render_frames(count=1)``)."""
        self.render_frames(count=1)

    def render_frames(self, poses=None, count: int | None = None) -> None:
        """Queue frames: one per pose of `poses` ((position, rotation) pairs or an (n, 7) array), or `count` frames of
        the camera's current pose.  world == 1: straight into ``framebuffer``.  world > 1: this rank's tiles of frame
        f are traced into the batch's compact buffer on render stream f mod F -- all frames of a call that belong to
        one batch by ONE library call (``rtgs_shard_frames``) -- and once the B frames of a batch are queued it is
        exchanged and landed on the caller's stream while the render streams go on with the next frames:
        ``framebuffer`` / ``landed_frames()`` are complete in the caller's stream order.  (Without a HIP backend --
        the CPU tests' stand-in -- or with RTGS_RENDER_STREAM=0: one stream, frame by frame.)"""
        b = self.backend
        if poses is not None:
            poses = _pose_rows(poses)
        n = len(poses) if poses is not None else int(1 if count is None else count)
        if n <= 0:
            return
        if self._comm is None:
            cam = getattr(b, "camera", None)
            for i in range(n):
                if poses is not None and cam is not None:
                    from .utils.types import vec3, vec4
                    cam.position, cam.rotation = vec3(poses[i, :3]), vec4(poses[i, 3:])
                self._render_one_serial()
            return
        import ctypes
        cm, nat = self._comm, b.nat
        B, F = self.gather_batch, self.frames_in_flight
        key = self._scene_key()
        stale = [s for s in range(F) if cm["seen"][s] != key]
        if stale:
            # scene uploads / edits queued on the caller's stream (rare: torch objects are fine here)
            cur_s = nat.current_stream(b.device.index)
            for s in stale:
                cm["render"][s].wait_stream(cur_s)
                cm["seen"][s] = key
        cur = nat.stream()                     # raw handle of the caller's stream
        cams, total = b.camera_array(poses, n)
        done = 0
        while done < n:
            f = self._frame
            q, j0 = divmod(f, B)
            m = min(n - done, B - j0)
            slot = q % self._slots
            out = self.compact[slot]
            base, stride = out.data_ptr(), out.stride(0) * 4
            outs = (ctypes.c_void_p * m)(*[(base + self._batch_position(j0 + i) * stride) if self.local_tiles else None
                                           for i in range(m)])
            nat.check(cm["lib"].rtgs_shard_frames(
                b.scene_native(), cams.ctypes.data + done * cams.itemsize, m, cm["params"], outs, cm["att_c"], cm["start_c"],
                cm["ws_c"], cm["ws"][0].numel(), cm["streams_c"], F, cm["turn"] % F, cm["sent"][slot], cm["done_c"][slot],
                cur, 1), "rtgs_shard_frames")
            # (the library deals the call's launches -- runs of up to frames_per_launch frames -- over the streams in turn)
            cm["turn"] += -(-m // self.frames_per_launch)
            self._frame += m
            done += m
            if j0 + m == B:
                self._exchange(slot, B, cur, q)

    def _render_one_serial(self) -> None:
        """One frame on the caller's stream (no render streams: the CPU tests' stand-in backend, RTGS_RENDER_STREAM=0)."""
        b = self.backend
        if self.world == 1:
            b.render(self._fbs[0], 0, 1, compact=False)
            return
        f = self._frame
        self._frame += 1
        if self.landing == "rotate":
            B = self.gather_batch
            q, j = divmod(f, B)
            slot = q % self._slots
            if self.local_tiles:
                out = self.compact[slot][self._batch_position(j)]
                if self._shard is not None:
                    b.render(out, self.rank, self.world, compact=True, shard=self._shard)
                else:
                    b.render(out, self.rank, self.world, compact=True)
            if j == B - 1:
                self._exchange_rotate_serial(slot, B, q)
            return
        k = f % self._slots
        if self.local_tiles:
            if self._shard is not None:
                b.render(self.compact[k], self.rank, self.world, compact=True, shard=self._shard)
            else:
                b.render(self.compact[k], self.rank, self.world, compact=True)
        if self.partition.arithmetic:
            work = self._gather(self.compact[k], gather_list=self._glist[k] if self._glist else None, dst=self.dst,
                                group=self.group, async_op=True)
        else:
            work = self._p2p(self.compact[k], self.landing_bufs[k] if self._fbs else None)
        self._pending.append((work, k))
        while len(self._pending) >= max(self.frames_in_flight, 2):
            self._complete(self._pending.pop(0))

    def _p2p(self, mine, landing):
        """The exchange step of an unequal deal: every rank sends its shard's own bytes to the destination, which
        receives each peer's into a buffer of that peer's size (what a gather is made of, without the padding to
        equal sizes).  Returns the works (complete in the current stream's order once waited for)."""
        if self._p2p_fn is not None:
            return self._p2p_fn(mine, landing)
        peers = [r for r in range(self.world) if r != self.rank and self.partition.counts[r]]
        if mine.is_cuda and dist.get_backend(self.group) == "gloo":
            # rehearsal only (gloo's send / recv take host pointers: a device tensor has to be staged by hand --
            # its gather does that by itself); RCCL moves device memory directly
            if self.rank == self.dst:
                host = {r: torch.empty(landing[r].shape, dtype=landing[r].dtype).pin_memory() for r in peers}
                works = [dist.irecv(host[r], r, group=self.group) for r in peers]
                return [_StagedRecv(w, host[r], landing[r]) for w, r in zip(works, peers)]
            return [dist.isend(mine.cpu(), self.dst, group=self.group)] if self.local_tiles else []
        ops = []
        if self.rank == self.dst:
            ops = [dist.P2POp(dist.irecv, landing[r], r, self.group) for r in peers]
        elif self.local_tiles:
            ops = [dist.P2POp(dist.isend, mine, self.dst, self.group)]
        return dist.batch_isend_irecv(ops) if ops else []

    def _all_to_all_batch(self, slot: int, g: int) -> None:
        """The rotating roots' exchange of batch slot `slot`: block d of this rank's send buffer (its shards of the
        batch's frames d, d + world, ...) goes to rank d; block s of the landing buffer comes from rank s."""
        send, recv = self.compact[slot], self.gathered[g]
        if self.partition.arithmetic:
            self._a2a(recv, send, None, None, group=self.group)
        else:
            self._a2a(recv[:sum(self._split_out)], send.reshape(-1), self._split_out, self._split_in, group=self.group)

    def _exchange(self, slot: int, n_frames: int, cur: int, batch: int) -> None:
        """Exchange batch slot `slot` (batch number `batch`) on the caller's stream (it already waits for the batch's
        traces) and land its first n_frames frames."""
        import ctypes
        cm, b = self._comm, self.backend
        g = cm["g"]
        cm["g"] = g ^ 1
        land = self._lands_here(n_frames)
        N, k = self.world, self.per_dest
        if self.landing == "rotate":
            self._all_to_all_batch(slot, g)
            if self.partition.arithmetic:
                b.nat.check(cm["lib"].rtgs_frames_land(
                    self.gathered[g].data_ptr(), N, self.tiles_per_rank, k, land, b.width, b.height, cm["images"], cur,
                    cm["sent"][slot]), "rtgs_frames_land")
            else:
                base = self.gathered[g].data_ptr()
                bufs = (ctypes.c_void_p * N)(*[(base + 4 * self._recv_off[r]) if self.partition.counts[r] else None
                                               for r in range(N)])
                b.nat.check(cm["lib"].rtgs_frames_land_map(
                    bufs, self._counts_c, N, b.nat.ptr(self._tile_map), k, land, b.width, b.height, cm["images"], cur,
                    cm["sent"][slot]), "rtgs_frames_land_map")
            self._note_landed(batch, land)
            return
        if not self.partition.arithmetic:
            for w in self._p2p(self.compact[slot], self.landing_bufs[g] if self._fbs else None):
                w.wait()                      # (NCCL: orders the caller's stream behind the transfer, no host wait)
            bufs = None
            if land:
                bufs = (ctypes.c_void_p * N)(*[
                    (self.compact[slot] if r == self.rank else self.landing_bufs[g][r]).data_ptr() if self.partition.counts[r] else None
                    for r in range(N)])
            b.nat.check(cm["lib"].rtgs_frames_land_map(
                bufs, self._counts_c, N, b.nat.ptr(self._tile_map), self.gather_batch, land, b.width, b.height,
                cm["images"], cur, cm["sent"][slot]), "rtgs_frames_land_map")
            self._note_landed(batch, land)
            return
        self._gather(self.compact[slot], gather_list=self._glist[g] if self._glist else None, dst=self.dst,
                     group=self.group, async_op=False)
        b.nat.check(cm["lib"].rtgs_frames_land(
            self.gathered[g].data_ptr() if self.gathered else None, N, self.tiles_per_rank,
            self.gather_batch, land, b.width, b.height, cm["images"], cur, cm["sent"][slot]), "rtgs_frames_land")
        self._note_landed(batch, land)

    def _exchange_rotate_serial(self, slot: int, n_frames: int, batch: int) -> None:
        """The rotating roots' exchange and landing without render streams (stand-in backend, RTGS_RENDER_STREAM=0)."""
        b = self.backend
        g = batch & 1
        self._all_to_all_batch(slot, g)
        land = self._lands_here(n_frames)
        N, k, T = self.world, self.per_dest, self.TILE
        for i in range(land):
            if self.partition.arithmetic:
                # gathered[g] = [source][k][tiles][T][T][3]: frame i's shards are [:, i]
                view = self.gathered[g].reshape(N, k, self.tiles_per_rank, T, T, 3)[:, i]
                b.untile(view.contiguous() if not view.is_contiguous() else view, N, self.tiles_per_rank, self._fbs[i])
            else:
                flat, cnt = self.gathered[g], self.partition.counts
                bufs = [flat[self._recv_off[r]:self._recv_off[r] + self._split_out[r]].reshape(k, cnt[r], T, T, 3)[i]
                        if cnt[r] else None for r in range(N)]
                b.untile_map(bufs, cnt, self._tile_map, self._fbs[i])
        self._note_landed(batch, land)

    def _complete(self, pending) -> None:
        work, k = pending
        if isinstance(work, list):
            for w in work:
                w.wait()
            if self._fbs:
                bufs = [(self.compact[k] if r == self.rank else self.landing_bufs[k][r]) if self.partition.counts[r] else None
                        for r in range(self.world)]
                self.backend.untile_map(bufs, self.partition.counts, self._tile_map, self._fbs[0])
            return
        work.wait()
        if self._fbs:
            self.backend.untile(self.gathered[k], self.world, self.tiles_per_rank, self._fbs[0])

    def finish(self) -> None:
        """Drain the pipeline: after this ``framebuffer`` (``landed_frames()``) holds the last frame(s) (in the
        caller's stream order) and the render streams have nothing queued that the caller's stream
        has not waited for -- buffers may be released, the scene re-committed.  A partial batch is
        exchanged as it is (every rank calls finish() after the same number of frames)."""
        while self._pending:
            self._complete(self._pending.pop(0))
        B = self.gather_batch
        q, j = divmod(self._frame, B)
        if self._comm is not None:
            b = self.backend
            if j:
                self._exchange(q % self._slots, j, b.nat.stream(), q)
                self._frame = (q + 1) * B          # the next frame opens a new batch
            cur = b.nat.current_stream(b.device.index)
            for rs in self._comm["render"]:
                cur.wait_stream(rs)
        elif self.world > 1 and self.landing == "rotate" and j:
            self._exchange_rotate_serial(q % self._slots, j, q)
            self._frame = (q + 1) * B

    # ---- pieces, for measurement -----------------------------------------------------------
    def project_only(self) -> None:
        self.backend.project()

    def trace_only(self) -> None:
        """The k_trace launch of this rank's shard alone (projection already in the workspace)."""
        b = self.backend
        if self.world == 1:
            b.trace(self.framebuffer, 0, 1, compact=False, projected=True)
        elif self.local_tiles:
            out = self.compact[0]
            if out.dim() == 5:           # a batch buffer: its first frame
                out = out[0]
            b.trace(out, self.rank, self.world, compact=True, projected=True, shard=self._shard)
