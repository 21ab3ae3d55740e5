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
        need = nat.lib().rtgs_workspace_bytes(scene.num_gaussians)
        self.workspace = torch.empty(need, dtype=torch.uint8, device=self.device)
        self._orders = {}
        self._state = {}
        # Optional hipGraph replay of project + trace (RTGS_GRAPH=1).  OFF by default: measured on
        # MI355X / ROCm 7.2 (tools/graph_vs_direct.py) a replay costs 12 us more per frame than the
        # two direct launches and holds the host until the graph has run (c3 1/8 shard: 70.9 vs
        # 58.1 us per frame; direct launches cost the host 13.5 us and overlap the GPU)
        self.use_graph = os.environ.get("RTGS_GRAPH", "0") == "1"
        self._graphs = {}
        # Optional (RTGS_PIPELINE=1): frame f+1's projection on a second stream, into a second
        # workspace, while frame f is still being traced.  OFF by default: measured on MI355X / ROCm
        # 7.2 the two cross-stream event waits per frame cost more than the overlap returns (c3
        # 0.1553 -> 0.1618 ms per frame, c5 1.345 -> 1.382 with the host queueing 200 frames ahead)
        self.pipeline = os.environ.get("RTGS_PIPELINE", "0") == "1"
        self._pipe = None

    def tile_order(self, tile_first: int, tile_step: int):
        key = (tile_first, tile_step)
        if key not in self._orders:
            o = cost_sorted_order(self.width, self.height, self.tile, tile_first, tile_step,
                                  "centre" if self.tile_order_mode == "centre" else "wedge")
            self._orders[key] = torch.from_numpy(o).to(self.device)
        return self._orders[key]

    def alloc(self, *shape):
        return torch.zeros(shape, dtype=torch.float32, device=self.device)

    def project(self):
        nat = self.nat
        nat.check(nat.lib().rtgs_project(self.scene.native(), self.camera.native(),
                                         nat.ptr(self.workspace), self.workspace.numel(),
                                         nat.stream()), "rtgs_project")

    def trace(self, out, tile_first: int, tile_step: int, compact: bool, projected: bool = False,
              split=None, workspace=None, stream=None):
        """split: None = the library picks the kernel form by launch size; True / False force the
        two-waves-per-tile / one-wave-per-tile form (same pixels either way).  stream: a
        torch.cuda.Stream to launch on instead of the current one (the C ABI takes the stream as
        an argument: no need to make it current)."""
        nat = self.nat
        ws = self.workspace if workspace is None else workspace
        st = nat.stream() if stream is None else stream.cuda_stream
        order = self.tile_order(tile_first, tile_step) if self.use_tile_order else None
        flags = nat.TRACE_COMPACT if compact else 0
        if split is not None:
            flags |= nat.TRACE_SPLIT_ON if split else nat.TRACE_SPLIT_OFF
        params = nat.RtgsTraceParams(self.depth, self.t_cutoff, tile_first, tile_step, flags,
                                     nat.ptr(order))
        att = start = None
        if self.depth > nat.MAX_DEPTH:
            # a chain of launches through per-pixel state (rtgs_hip.h, RtgsTraceParams.depth): one
            # (attenuation, ray.start) pair per workspace, i.e. per frame in flight
            att, start = self._state.setdefault(ws.data_ptr(), (self.alloc(self.width, self.height),
                                                                 self.alloc(self.width, self.height)))
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

    def render(self, out, tile_first: int, tile_step: int, compact: bool, stream=None, workspace=None):
        """project + trace of one shard into `out`: two direct launches on `stream` (default: the
        current stream) with `workspace` (default: the backend's own), or -- opt-in, current stream
        only -- a graph replay / the pipelined form."""
        if stream is None and self.use_graph:
            self.frame_graph(out, tile_first, tile_step, compact).replay()
        elif stream is None and self.pipeline:
            self._render_pipelined(out, tile_first, tile_step, compact)
        else:
            self.trace(out, tile_first, tile_step, compact, stream=stream, workspace=workspace)

    def _render_pipelined(self, out, tile_first: int, tile_step: int, compact: bool):
        """project on a side stream into workspace f & 1, trace on the caller's stream once that
        projection is done.  A frame's own latency is what it was (its trace waits for its
        projection); a host that queues frames ahead of the GPU -- a turntable, bench.py, the
        tile-parallel loop -- gets frame f+1's k_project / k_bin executed in the shadow of frame f's
        k_trace, whose last workgroups leave most of the chip idle.  The pose and the scene pointers
        are read when the frame is queued, as without the pipeline."""
        nat = self.nat
        if self._pipe is None:
            self._pipe = dict(stream=torch.cuda.Stream(device=self.device), turn=0,
                              ws=[self.workspace, torch.empty_like(self.workspace)],
                              projected=[torch.cuda.Event(), torch.cuda.Event()],
                              traced=[torch.cuda.Event(), torch.cuda.Event()])
        pp = self._pipe
        k = pp["turn"] & 1
        pp["turn"] += 1
        main = torch.cuda.current_stream(self.device)
        side = pp["stream"]
        ws = pp["ws"][k]
        side.wait_event(pp["traced"][k])       # the trace two frames back was the last reader of ws
        if pp.get("scene_version") != self.scene.version:
            side.wait_stream(main)             # scene uploads / edits queued on the caller's stream
            pp["scene_version"] = self.scene.version
        with torch.cuda.stream(side):
            nat.check(nat.lib().rtgs_project_shard(self.scene.native(), self.camera.native(), tile_first,
                                                   tile_step, nat.ptr(ws), ws.numel(), nat.stream()),
                      "rtgs_project_shard")
            pp["projected"][k].record(side)
        main.wait_event(pp["projected"][k])
        self.trace(out, tile_first, tile_step, compact, projected=True, workspace=ws)
        pp["traced"][k].record(main)

    def untile(self, gathered, world: int, tiles_per_rank: int, image, stream=None):
        nat = self.nat
        st = nat.stream() if stream is None else stream.cuda_stream
        nat.check(nat.lib().rtgs_untile(nat.ptr(gathered), world, tiles_per_rank, self.width,
                                        self.height, nat.ptr(image), st), "rtgs_untile")


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
    throughput, not by one tile's latency.  Each frame still costs its render stream exactly one
    event record; the gather and the untile stay on the caller's stream, in frame order."""

    def __init__(self, scene=None, camera=None, depth: int = 16, rank: int = 0, world: int = 1,
                 t_cutoff: float = 0.0, dst: int = 0, group=None, backend=None, gather_fn=None,
                 frames_in_flight: int | None = None):
        self.backend = backend if backend is not None else HipBackend(scene, camera, depth, t_cutoff)
        self._gather = gather_fn if gather_fn is not None else dist.gather   # injectable for tests
        self.rank, self.world, self.dst, self.group = int(rank), int(world), int(dst), group
        b = self.backend
        self.TILE = int(b.tile)
        self.total_tiles = tiles_of(b.width, b.height, self.TILE)
        self.tiles_per_rank = (self.total_tiles + self.world - 1) // self.world
        self.local_tiles = local_tile_count(self.total_tiles, self.rank, self.world)
        self.framebuffer = b.alloc(b.width, b.height, 3) if (self.world == 1 or self.rank == self.dst) else None
        if frames_in_flight is None:
            frames_in_flight = int(os.environ.get("RTGS_FRAMES_IN_FLIGHT", "2"))
        if frames_in_flight < 1:
            raise ValueError("frames_in_flight must be >= 1")
        self.frames_in_flight = int(frames_in_flight) if self.world > 1 else 1
        self._pending = []          # one-stream form: (work, slot) of the gathers not yet completed
        self._frame = 0
        self._comm = None
        if self.world > 1:
            F = self.frames_in_flight
            streams = isinstance(b, HipBackend) and os.environ.get("RTGS_RENDER_STREAM", "1") != "0"
            # buffer ring.  Two-stream form: 2 F slots, so that a render stream never waits for the
            # gather of the frame it traced last (only for the one before that, long done);
            # one-stream form: at most F gathers are outstanding, F slots
            self._slots = 2 * F if streams else max(F, 2)
            shard = (self.tiles_per_rank, self.TILE, self.TILE, 3)
            self.compact = [b.alloc(*shard) for _ in range(self._slots)]
            n_g = (2 if streams else self._slots) if self.rank == self.dst else 0
            # (two-stream form: gather and untile of consecutive frames are serial on the caller's
            # stream, two landing buffers are plenty; one-stream form: one per outstanding gather)
            self.gathered = [b.alloc(self.world, *shard) for _ in range(n_g)]
            # the per-rank views torch.distributed.gather wants, made once (8 views cost ~8 us per frame)
            self._glist = [list(g.unbind(0)) for g in self.gathered]
            if streams:
                # F + 1 streams: the shards are rendered on streams of their own (handed to the C ABI
                # as an argument, never made current), the gather and the untile stay on the caller's
                # stream.  Neither side waits for the other's current frame -- only for the buffer it
                # is about to reuse -- and the collective can be issued in its blocking form, which
                # costs the host 16 us instead of the 28 us of async_op=True + wait()
                # (tools/host_overhead.py)
                self._comm = dict(render=[torch.cuda.Stream(device=b.device) for _ in range(F)],
                                  ws=[b.workspace] + [torch.empty_like(b.workspace) for _ in range(F - 1)],
                                  traced=[torch.cuda.Event() for _ in range(self._slots)],
                                  sent=[torch.cuda.Event() for _ in range(self._slots)],
                                  seen=[None] * F)

    # ---- frame loop ------------------------------------------------------------------------
    def _scene_key(self):
        """Changes when work queued on the CALLER's stream may have written what the render streams
        read: a re-commit, or an in-place edit of a streamed tensor (torch bumps ``_version``)."""
        sc = self.backend.scene
        sh = sc.sh
        return (sc.version, sc.position._version, sc.rotation._version, sc.scale._version,
                sc.color_opacity._version, -1 if sh is None else sh._version)

    def render_frame(self) -> None:
        """Queue one frame.  world == 1: straight into ``framebuffer``.  world > 1: trace this
        rank's tiles into a slot of the compact-buffer ring on render stream f mod F, gather them
        (and, on rank dst, untile) on the caller's stream while the render streams go on with the
        next frames: ``framebuffer`` is complete in the caller's stream order.  (Without a HIP
        backend -- the CPU tests' stand-in -- or with RTGS_RENDER_STREAM=0: one stream; start the
        gather, then complete the oldest outstanding frame once F are in flight.)"""
        b = self.backend
        if self.world == 1:
            b.render(self.framebuffer, 0, 1, compact=False)
            return
        f = self._frame
        self._frame += 1
        k = f % self._slots
        if self._comm is not None:
            cm = self._comm
            s = f % self.frames_in_flight
            rs = cm["render"][s]
            cur = b.nat.current_stream(b.device.index)    # the caller's stream
            key = self._scene_key()
            if cm["seen"][s] != key:
                rs.wait_stream(cur)                # scene uploads / edits queued on the caller's stream
                cm["seen"][s] = key
            # the gather 2 F frames back has read compact[k].  (Guarding the reuse from the host
            # instead -- event.synchronize() on a ring of 2 or 4 buffers -- keeps this barrier packet
            # off the render stream but blocks a host that runs ahead, every frame: 79 / 73 us per
            # frame against 67.)
            if not cm["sent"][k].query():          # usually long done: then the render stream is spared the packet
                rs.wait_event(cm["sent"][k])
            if self.local_tiles:
                b.render(self.compact[k], self.rank, self.world, compact=True, stream=rs, workspace=cm["ws"][s])
            cm["traced"][k].record(rs)
            cur.wait_event(cm["traced"][k])        # gather this frame's shard once it is traced
            g = f & 1
            self._gather(self.compact[k], gather_list=self._glist[g] if self._glist else None, dst=self.dst,
                         group=self.group, async_op=False)
            cm["sent"][k].record(cur)
            if self.rank == self.dst:
                b.untile(self.gathered[g], self.world, self.tiles_per_rank, self.framebuffer,
                         stream=cur)
            return
        if self.local_tiles:
            b.render(self.compact[k], self.rank, self.world, compact=True)
        work = self._gather(self.compact[k], gather_list=self._glist[k] if self._glist else None, dst=self.dst,
                            group=self.group, async_op=True)
        self._pending.append((work, k))
        while len(self._pending) >= max(self.frames_in_flight, 2):
            self._complete(self._pending.pop(0))

    def _complete(self, pending) -> None:
        work, k = pending
        work.wait()
        if self.rank == self.dst:
            self.backend.untile(self.gathered[k], self.world, self.tiles_per_rank, self.framebuffer)

    def finish(self) -> None:
        """Drain the pipeline: after this ``framebuffer`` on rank dst holds the last frame (in the
        caller's stream order) and the render streams have nothing queued that the caller's stream
        has not waited for -- buffers may be released, the scene re-committed."""
        while self._pending:
            self._complete(self._pending.pop(0))
        if self._comm is not None:
            cur = self.backend.nat.current_stream(self.backend.device.index)
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
            b.trace(self.compact[0], self.rank, self.world, compact=True, projected=True)
