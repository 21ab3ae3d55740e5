"""Tile-parallel rendering: one process per GPU, macro-tiles interleaved over ranks, RCCL gather.

The reference is single-process / single-device (no collective anywhere).  Pixels
are independent, so the frame shards naturally: the T x T macro-tiles (T = rtgs_tile_size())
are dealt round-robin
(tile t -> rank t % world), every rank holds the whole scene (one broadcast at load, Gaussians
are replicated) and renders its tiles into a compact [tiles_per_rank, T, T, 3] buffer, and ONE
exchange step -- a ``torch.distributed`` gather (backend "nccl" == RCCL over xGMI) of
W*H*12/world bytes per rank -- lands the shards on the destination rank, where ``rtgs_untile``
de-interleaves them into the (W, H, 3) framebuffer.  Every rank keeps F frames in flight
(``TileParallelRenderer``, default 2): frame f is traced on render stream f mod F while the gather of
the frames before it runs on the caller's stream.

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


def _side_stream(device, priority: str = "low"):
    """A stream for work that should fill the chip only where the caller's stream leaves it idle (the next frame's
    projection behind this frame's trace): the LOWEST priority the device offers (torch.cuda.Stream only reaches the
    normal and the high level), wrapped for torch.  priority="normal": a plain torch stream."""
    if priority != "low":
        return torch.cuda.Stream(device=device)
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    least, greatest = ctypes.c_int(0), ctypes.c_int(0)
    if hip.hipDeviceGetStreamPriorityRange(ctypes.byref(least), ctypes.byref(greatest)) != 0:
        return torch.cuda.Stream(device=device)
    handle = ctypes.c_void_p()
    if hip.hipStreamCreateWithPriority(ctypes.byref(handle), 1, least.value) != 0:      # 1 = hipStreamNonBlocking
        return torch.cuda.Stream(device=device)
    return torch.cuda.ExternalStream(handle.value, device=device)


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
        # Optional hipGraph replay of project + trace (RTGS_GRAPH=1).  OFF by default: measured on
        # MI355X / ROCm 7.2 (tools/graph_vs_direct.py) a replay costs 12 us more per frame than the
        # two direct launches and holds the host until the graph has run (c3 1/8 shard: 70.9 vs
        # 58.1 us per frame; direct launches cost the host 13.5 us and overlap the GPU)
        self.use_graph = os.environ.get("RTGS_GRAPH", "0") == "1"
        self._graphs = {}
        # Optional (RTGS_PIPELINE=1): frame f+1's projection on a second stream, into a second
        # workspace, while frame f is still being traced.  OFF by default: measured on MI355X / ROCm
        # 7.2 the two cross-stream event waits per frame cost more than the overlap returns -- also as ONE
        # library call with raw events (rtgs_frame_ahead, round 4): c3 0.1502 -> 0.1547 ms per frame,
        # c5 1.320 -> 1.344, c2 0.0485 -> 0.0541 with the host queueing 200 frames ahead
        self.pipeline = os.environ.get("RTGS_PIPELINE", "0") == "1"
        self._pipe = None

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
              split=None, workspace=None, stream=None, shard=None):
        """split: None = the library picks the kernel form by launch size; True / False force the
        two-waves-per-tile / one-wave-per-tile form (same pixels either way).  stream: a
        torch.cuda.Stream to launch on instead of the current one (the C ABI takes the stream as
        an argument: no need to make it current)."""
        nat = self.nat
        ws = self.workspace if workspace is None else workspace
        st = nat.stream() if stream is None else stream.cuda_stream
        flags = nat.TRACE_COMPACT if compact else 0
        if split is not None:
            flags |= nat.TRACE_SPLIT_ON if split else nat.TRACE_SPLIT_OFF
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

    # ---- hipGraph path ---------------------------------------------------------------------
    def frame_graph(self, out, tile_first: int, tile_step: int, compact: bool):
        """A captured hipGraph of [pose upload, rtgs_project, rtgs_trace] writing into `out`.  The
        camera pose travels through a device block (RtgsCamera.device_pose) that the graph refreshes
        from pinned host memory, so a replay renders the pose packed just before it."""
        key = (out.data_ptr(), tile_first, tile_step, compact)
        g = self._graphs.get(key)
        if g is None:
            g = _FrameGraph(self, out, tile_first, tile_step, compact)
            self._graphs[key] = g
        return g

    def render(self, out, tile_first: int, tile_step: int, compact: bool, stream=None, workspace=None, shard=None):
        """project + trace of one shard into `out`: two direct launches on `stream` (default: the
        current stream) with `workspace` (default: the backend's own), or -- opt-in, current stream
        only -- a graph replay / the pipelined form."""
        if stream is None and self.use_graph and shard is None:
            self.frame_graph(out, tile_first, tile_step, compact).replay()
        elif stream is None and self.pipeline and shard is None:
            self._render_pipelined(out, tile_first, tile_step, compact)
        else:
            self.trace(out, tile_first, tile_step, compact, stream=stream, workspace=workspace, shard=shard)

    def _render_pipelined(self, out, tile_first: int, tile_step: int, compact: bool):
        """project on a side stream into workspace f & 1, trace on the caller's stream once that projection is
        done: ONE library call (``rtgs_frame_ahead``: two stream waits, two event records, the launches).  A
        frame's own latency is what it was (its trace waits for its projection); a host that queues frames ahead
        of the GPU -- a turntable, bench.py -- gets frame f+1's k_project / k_bin (latency-bound, a few CUs)
        executed while frame f's k_trace fills the chip.  The pose and the scene pointers are read when the frame
        is queued, as without the pipeline."""
        import ctypes
        nat = self.nat
        lib = nat.lib()
        if self._pipe is None:
            def event():
                h = ctypes.c_void_p()
                nat.check(lib.rtgs_event_create(ctypes.byref(h)), "rtgs_event_create")
                return h.value
            self._pipe = dict(stream=_side_stream(self.device, os.environ.get("RTGS_PIPELINE_PRIO", "low")), turn=0,
                              ws=[self.workspace, torch.empty_like(self.workspace)],
                              projected=[event(), event()], traced=[event(), event()], params={})
        pp = self._pipe
        k = pp["turn"] & 1
        pp["turn"] += 1
        side = pp["stream"]
        key = self.scene.content_key()
        if pp.get("scene_key") != key:
            side.wait_stream(nat.current_stream(self.device.index))   # scene uploads / edits queued on the caller's stream
            pp["scene_key"] = key
        pk = (tile_first, tile_step, compact)
        if pk not in pp["params"]:
            order = self.tile_order(tile_first, tile_step) if self.use_tile_order else None
            pp["params"][pk] = (nat.RtgsTraceParams(self.depth, self.t_cutoff, tile_first, tile_step,
                                                    nat.TRACE_COMPACT if compact else 0, nat.ptr(order)), order)
        ws = pp["ws"][k]
        nat.check(lib.rtgs_frame_ahead(self.scene_native(), self.camera_native(), pp["params"][pk][0], nat.ptr(out), None, None,
                                       nat.ptr(ws), ws.numel(), side.cuda_stream, nat.stream(), pp["traced"][k],
                                       pp["projected"][k], pp["traced"][k]), "rtgs_frame_ahead")

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


class _FrameGraph:
    """One captured frame of one shard (see HipBackend.frame_graph): the graph holds rtgs_project +
    rtgs_trace; the pose block they read is refreshed before every replay by one async copy from a
    small ring of pinned host blocks, so the host never waits for the GPU."""

    RING = 4

    def __init__(self, backend: HipBackend, out, tile_first, tile_step, compact):
        nat = backend.nat
        self.b = backend
        nbytes = nat.lib().rtgs_pose_bytes()
        self.pose_host = [torch.empty(nbytes, dtype=torch.uint8).pin_memory() for _ in range(self.RING)]
        self.copied = [torch.cuda.Event() for _ in range(self.RING)]   # the copy out of block k is done
        self.pose_dev = torch.empty(nbytes, dtype=torch.uint8, device=backend.device)
        self.turn = 0
        self._upload()
        order = backend.tile_order(tile_first, tile_step) if backend.use_tile_order else None
        flags = nat.TRACE_COMPACT if compact else 0
        params = nat.RtgsTraceParams(backend.depth, backend.t_cutoff, tile_first, tile_step, flags, nat.ptr(order))
        scene = backend.scene.native()
        self.scene_version = backend.scene.version
        ncam = backend.camera.native(device_pose=self.pose_dev.data_ptr())
        self.graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(self.graph):
            nat.check(nat.lib().rtgs_trace(scene, ncam, params, nat.ptr(out), None, None, 1,
                                           nat.ptr(backend.workspace), backend.workspace.numel(), 0,
                                           nat.stream()), "rtgs_trace (capture)")
        self._keep = (order, params, scene, ncam)

    def _upload(self):
        nat = self.b.nat
        k = self.turn % self.RING
        self.turn += 1
        self.copied[k].synchronize()        # RING frames old: long complete
        nat.check(nat.lib().rtgs_pose_pack(self.b.camera.native(), self.pose_host[k].data_ptr()), "rtgs_pose_pack")
        self.pose_dev.copy_(self.pose_host[k], non_blocking=True)
        self.copied[k].record()

    def replay(self):
        if self.scene_version != self.b.scene.version:
            raise RuntimeError("the scene was re-committed after this frame graph was captured")
        self._upload()
        self.graph.replay()


class TileParallelRenderer:
    """Renders whole frames on `world` ranks; the image lands in ``framebuffer`` on rank `dst`.

    ``frames_in_flight`` (default 2, ``RTGS_FRAMES_IN_FLIGHT``): a 1/8 shard of a 1080p frame is ONE
    resident round of workgroups, so its trace takes as long as its heaviest tile's dependency chain
    (37 us at c3 whether the shard is 1/8 or 1/16 of the frame) and most of the chip idles behind it.
    With F frames in flight every rank alternates its frames over F render streams, each with its
    own workspace, so consecutive frames' traces overlap (two 1/8-shard traces: 43.5 us per pair
    against 74.3 one after the other, tools/concurrency_probe.py) and the loop is bound by
    throughput, not by one tile's latency.  Each frame costs its render stream exactly one event
    record; the gather and the untile stay on the caller's stream, in frame order.  The render side
    of a frame is ONE library call (``rtgs_shard_frame``: reuse wait, project, trace, event record,
    consumer wait), the landing side another (``rtgs_frames_land``: event record + untile).

    ``gather_batch`` (default 1, ``RTGS_GATHER_BATCH``): B consecutive frames' shards travel in one
    collective (each rank sends [B][tiles][T][T][3]) and land in a ring of B framebuffers, the
    newest being ``framebuffer``: the collective's fixed host and launch cost (16-26 us per call,
    more than a 1/8-shard trace of a 1080p frame) is paid once per B frames, at the price of B - 1
    frames of latency.  ``finish()`` flushes a partial batch."""

    def __init__(self, scene=None, camera=None, depth: int = 16, rank: int = 0, world: int = 1,
                 t_cutoff: float = 0.0, dst: int = 0, group=None, backend=None, gather_fn=None,
                 frames_in_flight: int | None = None, gather_batch: int | None = None, partition=None, p2p_fn=None):
        """``partition``: None / "equal" = the round-robin deal (equal shards, one ``dist.gather``); a ``Partition``
        or a sequence of per-rank shares = that deal (point-to-point exchange of each shard's own bytes); "auto" =
        ``calibrated_shares`` (every rank must construct its renderer at the same time: it times one shard trace,
        one frame trace and one exchange on this machine and the destination broadcasts the shares)."""
        self.backend = backend if backend is not None else HipBackend(scene, camera, depth, t_cutoff)
        self._gather = gather_fn if gather_fn is not None else dist.gather   # injectable for tests
        self._p2p_fn = p2p_fn                                                # likewise: (mine, landing) -> works
        self.rank, self.world, self.dst, self.group = int(rank), int(world), int(dst), group
        b = self.backend
        self.TILE = int(b.tile)
        self.total_tiles = tiles_of(b.width, b.height, self.TILE)
        self.calibration = None
        if isinstance(partition, str) and partition == "wedges":
            partition = Partition.wedges(b.width, b.height, self.TILE, self.world)
        if isinstance(partition, str) and partition == "auto":
            partition = None
            if self.world > 1:
                self.calibration = calibrated_shares(b, self.rank, self.world, self.dst, self.group)
                if not self.calibration["equal"]:
                    partition = self.calibration["shares"]
        if partition is None or (isinstance(partition, str) and partition == "equal"):
            partition = Partition.round_robin(self.total_tiles, self.world)
        elif not isinstance(partition, Partition):
            partition = Partition.weighted(self.total_tiles, partition)
        if partition.world != self.world or partition.total != self.total_tiles:
            raise ValueError("partition is of another frame / world size")
        _phase("deal", partition.name, "local tiles", partition.counts[self.rank])
        self.partition = partition
        self.tiles_per_rank = (self.total_tiles + self.world - 1) // self.world     # the arithmetic deal's padded shard
        self.local_tiles = partition.counts[self.rank]
        # any other deal: this rank's tile list (device), the frame's tile -> (rank, slot) map (destination)
        self._shard = self._tile_map = None
        if not partition.arithmetic:
            self._shard = (b.shard(partition.tiles[self.rank], region=partition.compact) if hasattr(b, "shard")
                           else dict(tiles=partition.tiles[self.rank]))
            if self.world == 1 or self.rank == self.dst:
                self._tile_map = b.to_device(partition.tile_map) if hasattr(b, "to_device") else partition.tile_map
        if frames_in_flight is None:
            frames_in_flight = int(os.environ.get("RTGS_FRAMES_IN_FLIGHT", "2"))
        if gather_batch is None:
            gather_batch = int(os.environ.get("RTGS_GATHER_BATCH", "1"))
        if frames_in_flight < 1 or gather_batch < 1:
            raise ValueError("frames_in_flight and gather_batch must be >= 1")
        streams = (self.world > 1 and isinstance(b, HipBackend) and
                   os.environ.get("RTGS_RENDER_STREAM", "1") != "0")
        self.frames_in_flight = int(frames_in_flight) if self.world > 1 else 1
        self.gather_batch = int(gather_batch) if streams else 1
        is_dst = self.world == 1 or self.rank == self.dst
        self._fbs = [b.alloc(b.width, b.height, 3) for _ in range(self.gather_batch)] if is_dst else []
        self._landed = 0            # which of _fbs holds the newest frame
        self._pending = []          # one-stream form: (work, slot) of the gathers not yet completed
        self._frame = 0
        self._comm = None
        if self.world > 1:
            F, B = self.frames_in_flight, self.gather_batch
            arith = partition.arithmetic
            shard = ((self.tiles_per_rank if arith else max(self.local_tiles, 1)), self.TILE, self.TILE, 3)
            peer = lambda r: (max(partition.counts[r], 1), self.TILE, self.TILE, 3)
            if streams:
                # batch-slot ring: deep enough that a render stream never waits for the gather of the
                # batch it traced into last (only for one S batches back, long done)
                self._slots = (2 * F + B - 1) // B + 1
                self.compact = [b.alloc(B, *shard) for _ in range(self._slots)]
                # gather and landing of consecutive batches are serial on the caller's stream: two
                # landing buffers are plenty
                if arith:
                    self.gathered = [b.alloc(self.world, B, *shard) for _ in range(2)] if is_dst else []
                else:       # one landing buffer per peer, sized for that peer's share
                    self.landing = [[None if r == self.rank else b.alloc(B, *peer(r)) for r in range(self.world)]
                                    for _ in range(2)] if is_dst else []
                    self.gathered = []
            else:
                # one-stream form: at most F gathers outstanding, one slot and landing buffer each
                self._slots = max(F, 2)
                self.compact = [b.alloc(*shard) for _ in range(self._slots)]
                if arith:
                    self.gathered = [b.alloc(self.world, *shard) for _ in range(self._slots)] if is_dst else []
                else:
                    self.landing = [[None if r == self.rank else b.alloc(*peer(r)) for r in range(self.world)]
                                    for _ in range(self._slots)] if is_dst else []
                    self.gathered = []
            # the per-rank views torch.distributed.gather wants, made once (8 views cost ~8 us per frame)
            self._glist = [list(g.unbind(0)) for g in self.gathered]
            if streams:
                self._comm = self._make_stream_state(F, B)

    @property
    def framebuffer(self):
        """(W, H, 3) image of the newest frame that has landed (rank dst; None elsewhere)."""
        return self._fbs[self._landed] if self._fbs else None

    def _make_stream_state(self, F: int, B: int):
        """F render streams (handed to the C ABI as arguments, never made current) with a workspace
        each; per frame slot a `traced` event, per batch slot a `sent` event (raw hipEvent_t handles:
        the library records and waits on them inside rtgs_shard_frame / rtgs_frames_land)."""
        import ctypes
        b = self.backend
        nat = b.nat
        lib = nat.lib()

        def event():
            h = ctypes.c_void_p()
            nat.check(lib.rtgs_event_create(ctypes.byref(h)), "rtgs_event_create")
            return h.value

        if self._shard is not None:
            order = self._shard["order"]
            params = b.shard_params(self._shard, nat.TRACE_COMPACT)
        else:
            order = b.tile_order(self.rank, self.world) if (b.use_tile_order and self.local_tiles) else None
            params = nat.RtgsTraceParams(b.depth, b.t_cutoff, self.rank, self.world, nat.TRACE_COMPACT, nat.ptr(order))
        ws = [b.workspace] + [torch.empty_like(b.workspace) for _ in range(F - 1)]
        state = [(None, None)] * F
        if b.depth > nat.MAX_DEPTH:      # chained launches need per-pixel state, one pair per frame in flight
            state = [(b.alloc(b.width, b.height), b.alloc(b.width, b.height)) for _ in range(F)]
        images = (ctypes.c_void_p * B)(*[fb.data_ptr() for fb in self._fbs]) if self._fbs else None
        self._counts_c = (ctypes.c_int32 * self.world)(*self.partition.counts)
        return dict(render=[torch.cuda.Stream(device=b.device) for _ in range(F)], ws=ws, state=state,
                    traced=[[event() for _ in range(B)] for _ in range(self._slots)],
                    sent=[event() for _ in range(self._slots)], seen=[None] * F,
                    params=params, order=order, images=images, g=0, lib=lib)

    def close(self) -> None:
        """Drain the render streams and release the events (also runs when the object is collected)."""
        cm, self._comm = self._comm, None
        if cm is not None:
            for rs in cm["render"]:
                rs.synchronize()
            for ev in [e for row in cm["traced"] for e in row] + cm["sent"]:
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
        """Queue one frame.  world == 1: straight into ``framebuffer``.  world > 1: trace this
        rank's tiles into a slot of the compact-buffer ring on render stream f mod F; once the B
        frames of a batch are queued, gather them (and, on rank dst, untile) on the caller's stream
        while the render streams go on with the next frames: ``framebuffer`` is complete in the
        caller's stream order.  (Without a HIP backend -- the CPU tests' stand-in -- or with
        RTGS_RENDER_STREAM=0: one stream; start the gather, then complete the oldest outstanding
        frame once F are in flight.)"""
        b = self.backend
        if self.world == 1:
            b.render(self._fbs[0], 0, 1, compact=False)
            return
        f = self._frame
        self._frame += 1
        if self._comm is not None:
            cm, nat = self._comm, b.nat
            B = self.gather_batch
            q, j = divmod(f, B)
            slot = q % self._slots
            s = f % self.frames_in_flight
            rs = cm["render"][s]
            key = self._scene_key()
            if cm["seen"][s] != key:
                # scene uploads / edits queued on the caller's stream (rare: torch objects are fine here)
                rs.wait_stream(nat.current_stream(b.device.index))
                cm["seen"][s] = key
            cur = nat.stream()                     # raw handle of the caller's stream
            ws = cm["ws"][s]
            att, start = cm["state"][s]
            out = self.compact[slot]
            nat.check(cm["lib"].rtgs_shard_frame(
                b.scene_native(), b.camera_native(), cm["params"],
                out.data_ptr() + j * out.stride(0) * 4, nat.ptr(att), nat.ptr(start),
                ws.data_ptr(), ws.numel(), rs.cuda_stream, cm["sent"][slot], cm["traced"][slot][j], cur, 1),
                "rtgs_shard_frame")
            if j == B - 1:
                self._exchange(slot, B, cur)
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
            work = self._p2p(self.compact[k], self.landing[k] if self._fbs else None)
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

    def _exchange(self, slot: int, n_frames: int, cur: int) -> None:
        """Gather batch slot `slot` on the caller's stream (it already waits for the batch's traces)
        and land its first n_frames frames."""
        cm, b = self._comm, self.backend
        g = cm["g"]
        cm["g"] = g ^ 1
        if not self.partition.arithmetic:
            import ctypes
            for w in self._p2p(self.compact[slot], self.landing[g] if self._fbs else None):
                w.wait()                      # (NCCL: orders the caller's stream behind the transfer, no host wait)
            land = n_frames if self._fbs else 0
            bufs = None
            if land:
                bufs = (ctypes.c_void_p * self.world)(*[
                    (self.compact[slot] if r == self.rank else self.landing[g][r]).data_ptr() if self.partition.counts[r] else None
                    for r in range(self.world)])
            b.nat.check(cm["lib"].rtgs_frames_land_map(
                bufs, self._counts_c, self.world, b.nat.ptr(self._tile_map), self.gather_batch, land, b.width, b.height,
                cm["images"], cur, cm["sent"][slot]), "rtgs_frames_land_map")
            if land:
                self._landed = land - 1
            return
        self._gather(self.compact[slot], gather_list=self._glist[g] if self._glist else None, dst=self.dst,
                     group=self.group, async_op=False)
        land = n_frames if self._fbs else 0
        b.nat.check(cm["lib"].rtgs_frames_land(
            self.gathered[g].data_ptr() if self.gathered else None, self.world, self.tiles_per_rank,
            self.gather_batch, land, b.width, b.height, cm["images"], cur, cm["sent"][slot]), "rtgs_frames_land")
        if land:
            self._landed = land - 1

    def _complete(self, pending) -> None:
        work, k = pending
        if isinstance(work, list):
            for w in work:
                w.wait()
            if self._fbs:
                bufs = [(self.compact[k] if r == self.rank else self.landing[k][r]) if self.partition.counts[r] else None
                        for r in range(self.world)]
                self.backend.untile_map(bufs, self.partition.counts, self._tile_map, self._fbs[0])
            return
        work.wait()
        if self._fbs:
            self.backend.untile(self.gathered[k], self.world, self.tiles_per_rank, self._fbs[0])

    def finish(self) -> None:
        """Drain the pipeline: after this ``framebuffer`` on rank dst holds the last frame (in the
        caller's stream order) and the render streams have nothing queued that the caller's stream
        has not waited for -- buffers may be released, the scene re-committed.  A partial batch is
        exchanged as it is (every rank calls finish() after the same number of frames)."""
        while self._pending:
            self._complete(self._pending.pop(0))
        if self._comm is not None:
            b, B = self.backend, self.gather_batch
            q, j = divmod(self._frame, B)
            if j:
                self._exchange(q % self._slots, j, b.nat.stream())
                self._frame = (q + 1) * B          # the next frame opens a new batch
            cur = b.nat.current_stream(b.device.index)
            for rs in self._comm["render"]:
                cur.wait_stream(rs)

    # ---- pieces, for measurement -----------------------------------------------------------
    def project_only(self) -> None:
        self.backend.project()

    def trace_only(self) -> None:
        """The k_trace launch of this rank's shard alone (projection already in the workspace)."""
        b = self.backend
        if self.world == 1:
            b.trace(self.framebuffer, 0, 1, compact=False, projected=True)
        elif self.local_tiles:
            b.trace(self.compact[0], self.rank, self.world, compact=True, projected=True, shard=self._shard)
