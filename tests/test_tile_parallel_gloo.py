"""The N > 1 path on CPU: world_size-2 (and 3) gloo processes run rtgs.parallel's real frame
loop (round-robin tile deal, double-buffered async gather, de-interleave) with a stand-in
backend whose "render" is the fp64 oracle and whose untile is written here in NumPy.  The HIP
backend itself is covered by the -m gpu tests (sharded == whole, rtgs_untile)."""

import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
W, H, TILE = 80, 72, 32     # 3 x 3 stand-in macro-tiles, ragged right/top edges


def _oracle_image():
    for p in (os.path.join(ROOT, "rt-gaussian-splat-renderer_amd"), ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    from oracle import c_oracle
    from rtgs import synthetic
    arrays = synthetic.make_scene(200, W / H, 5)
    cam = synthetic.bench_camera(W, H)
    img = c_oracle.render(arrays, cam, 16)["sample"].reshape(W, H, 3)
    return torch.from_numpy(img.astype(np.float32))


class OracleBackend:
    """Same surface as rtgs.parallel.HipBackend, CPU tensors; frames differ by a scale factor
    so that a stale or swapped double buffer would be caught."""

    def __init__(self, image):
        self.width, self.height, self.tile, self.image, self.frame = W, H, TILE, image, 0

    def alloc(self, *shape):
        return torch.zeros(shape, dtype=torch.float32)

    def project(self):
        pass

    def trace(self, out, tile_first, tile_step, compact, projected=False, shard=None):
        self.frame += 1
        img = self.image * float(self.frame)
        if not compact:
            out.copy_(img)
            return
        tx = (W + TILE - 1) // TILE
        total = tx * ((H + TILE - 1) // TILE)
        out.zero_()
        tiles = range(tile_first, total, tile_step) if shard is None else shard["tiles"].tolist()
        for local, t in enumerate(tiles):
            i0, j0 = (t % tx) * TILE, (t // tx) * TILE
            blk = img[i0:i0 + TILE, j0:j0 + TILE]
            out[local, :blk.shape[0], :blk.shape[1]] = blk

    def render(self, out, tile_first, tile_step, compact, shard=None):
        self.trace(out, tile_first, tile_step, compact, shard=shard)

    def untile_map(self, bufs, counts, tile_map, image):
        tx = (W + TILE - 1) // TILE
        for t in range(tx * ((H + TILE - 1) // TILE)):
            i0, j0 = (t % tx) * TILE, (t // tx) * TILE
            w, h = min(TILE, W - i0), min(TILE, H - j0)
            m = int(tile_map[t])
            image[i0:i0 + w, j0:j0 + h] = bufs[m >> 24][m & 0xFFFFFF, :w, :h]

    def untile(self, gathered, world, tiles_per_rank, image):
        tx = (W + TILE - 1) // TILE
        for t in range(tx * ((H + TILE - 1) // TILE)):
            i0, j0 = (t % tx) * TILE, (t // tx) * TILE
            w, h = min(TILE, W - i0), min(TILE, H - j0)
            image[i0:i0 + w, j0:j0 + h] = gathered[t % world, t // world, :w, :h]


def _worker(rank, world, port, frames, q, in_flight=2, shares=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        image = _oracle_image()
        from rtgs.parallel import Partition, TileParallelRenderer, local_tile_count, tiles_of
        be = OracleBackend(image)
        r = TileParallelRenderer(backend=be, rank=rank, world=world, frames_in_flight=in_flight, partition=shares)
        f0 = be.frame                                # ("auto" has rendered calibration frames, the same number on every rank)
        assert r.frames_in_flight == in_flight and len(r.compact) == max(in_flight, 2)
        assert r.total_tiles == tiles_of(W, H, TILE) == 9
        if shares is None:
            assert r.partition.arithmetic and r.local_tiles == local_tile_count(9, rank, world)
        elif shares == "wedges":
            assert r.partition.compact and not r.partition.arithmetic and sum(r.partition.counts) == 9
        elif shares == "auto":
            assert r.calibration is not None and abs(sum(r.calibration["shares"]) - 1) < 1e-9
            assert sum(r.partition.counts) == 9
        else:
            want = Partition.weighted(9, shares)
            assert not r.partition.arithmetic and r.partition.counts == want.counts and r.local_tiles == want.counts[rank]
        for k in range(frames):
            r.render_frame()
            # at most F frames are outstanding after a call (F - 1 once the oldest has been completed)
            assert len(r._pending) == min(k + 1, max(in_flight, 2) - 1)
        r.finish()
        assert not r._pending
        if rank == 0:
            q.put(float((r.framebuffer - image * float(f0 + frames)).abs().max()))
    finally:
        dist.destroy_process_group()


def _rotate_worker(rank, world, port, frames, q, batch, shares, poses_per_call):
    """landing="rotate": of every batch of B = k world frames the frames [d k, (d + 1) k) land on rank d, one
    all_to_all_single per batch; EVERY landed frame is checked on the rank that holds it, after every batch and
    after the partial last one."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        image = _oracle_image()
        from rtgs.parallel import TileParallelRenderer
        be = OracleBackend(image)
        r = TileParallelRenderer(backend=be, rank=rank, world=world, partition=shares, landing="rotate", gather_batch=batch)
        B = r.gather_batch
        assert B % world == 0 and B >= batch and r.per_dest == B // world and len(r._fbs) == r.per_dest
        seen = {}

        def check():
            for f, img in r.landed_frames():
                assert r.destination(f) == rank == (f % B) // (B // world)
                if f not in seen:            # (a batch that lands nothing here leaves the previous one's frames named)
                    seen[f] = float((img - image * float(f + 1)).abs().max())

        done = 0
        while done < frames:
            n = min(poses_per_call, frames - done)
            before = done // B
            r.render_frames(count=n)
            done += n
            if done // B != before:
                check()
            # an empty share renders nothing: keep the stand-in's frame counter in step with the peers'
            be.frame = done
        r.finish()
        if frames % B:
            check()
        q.put((rank, seen))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,frames,batch,shares,per_call", [
    (2, 4, 2, None, 1), (2, 7, 4, None, 3), (3, 10, 3, None, 1), (3, 7, 6, None, 7), (4, 9, 4, None, 2),
    (2, 5, 2, (3, 1), 1), (3, 8, 3, (5, 0, 1), 2), (3, 4, 6, "wedges", 4), (4, 6, 8, (1, 2, 0, 4), 6)])
def test_rotating_roots_land_every_frame_on_its_rank(world, frames, batch, shares, per_call):
    # world-2/3/4, equal / unequal / empty shares, several frames per call, a partial last batch:
    # every frame of the LAST complete-or-partial batch and of every batch before it must be bit-exact on its rank
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    mp.spawn(_rotate_worker, args=(world, _free_port(), frames, q, batch, shares, per_call), nprocs=world, join=True)
    got = dict(q.get(timeout=30) for _ in range(world))
    B = (batch + world - 1) // world * world
    landed = {}
    for rank, seen in got.items():
        for f, err in seen.items():
            assert (f % B) // (B // world) == rank and err == 0.0, (rank, f, err)
            landed[f] = rank
    # batches are only looked at when a call crosses their end (the ring of k framebuffers is overwritten by the next
    # batch): the last batch -- complete or partial -- is always seen in full
    last = range((frames - 1) // B * B, frames)
    assert set(last) <= set(landed), (sorted(landed), list(last))
    if per_call <= B:                # (a call then ends at most one batch: every batch was looked at)
        assert set(landed) == set(range(frames))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,frames,in_flight", [(2, 1, 2), (2, 4, 2), (3, 3, 2), (2, 7, 3), (3, 5, 3), (2, 2, 1)])
def test_sharded_frames_reassemble_on_rank0(world, frames, in_flight):
    # every frame differs (OracleBackend scales by the frame number): the framebuffer must hold the
    # LAST one whatever the ring position it ends on, with 1, 2 and 3 frames in flight
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    mp.spawn(_worker, args=(world, _free_port(), frames, q, in_flight), nprocs=world, join=True)
    assert q.get(timeout=30) == 0.0


def test_single_rank_needs_no_process_group():
    image = _oracle_image()
    from rtgs.parallel import TileParallelRenderer
    r = TileParallelRenderer(backend=OracleBackend(image), rank=0, world=1)
    r.render_frame()
    r.finish()
    assert torch.equal(r.framebuffer, image)


def test_tile_deal_covers_every_tile_once():
    from rtgs import _native as nat
    from rtgs.parallel import local_tile_count, tiles_of
    lib = nat.lib()
    for (w, h) in [(1920, 1080), (3840, 2160), (800, 600), (33, 1)]:
        total = tiles_of(w, h, nat.tile_size())
        assert total == lib.rtgs_num_tiles(w, h)
        for world in (1, 2, 3, 8):
            counts = [local_tile_count(total, r, world) for r in range(world)]
            assert counts == [lib.rtgs_num_local_tiles(w, h, r, world) for r in range(world)]
            assert sum(counts) == total and max(counts) - min(counts) <= 1


def test_launch_orders_are_permutations_and_give_every_xcd_one_wedge():
    """Launch orders only reschedule workgroups: each must name every local tile exactly once, for
    whole frames and shards; in the wedge order the tiles at positions 8 i + x (workgroups the
    round-robin dispatch hands to XCD x) must come from one angular sector of the image."""
    import numpy as np
    from rtgs import _native as nat
    from rtgs.parallel import cost_sorted_order, local_tile_count, tiles_of
    T = nat.tile_size()
    for (w, h) in [(1920, 1080), (3840, 2160), (800, 600), (33, 1), (160, 160)]:
        total = tiles_of(w, h, T)
        for world in (1, 2, 8):
            for rank in {0, world - 1}:
                n = local_tile_count(total, rank, world)
                for mode in ("centre", "wedge"):
                    o = cost_sorted_order(w, h, T, rank, world, mode)
                    assert o.dtype == np.int32 and sorted(o.tolist()) == list(range(n)), (w, h, world, rank, mode)
    w, h = 1920, 1080
    o = cost_sorted_order(w, h, T, 0, 1, "wedge")
    tx = (w + T - 1) // T
    ang = np.arctan2((o // tx + 0.5) * T - h / 2, (o % tx + 0.5) * T - w / 2)
    n8 = len(o) // 8 * 8
    spans = sorted((ang[x:n8:8].min(), ang[x:n8:8].max()) for x in range(8))
    for (lo0, hi0), (lo1, hi1) in zip(spans, spans[1:]):
        assert hi0 <= lo1                                   # sectors do not overlap
    # most central first inside a wedge
    m = np.maximum(np.abs((o % tx + 0.5) * T - w / 2) / (w / 2), np.abs((o // tx + 0.5) * T - h / 2) / (h / 2))
    assert np.all(np.diff(m[0:n8:8]) >= 0)


@pytest.mark.parametrize("world,frames,in_flight,shares", [(2, 3, 2, (3, 1)), (3, 4, 2, (5, 1, 3)), (3, 2, 3, (5, 0, 1)),
                                                             (2, 5, 1, (1, 7)), (2, 3, 2, "auto"), (3, 3, 2, "wedges")])
def test_unequal_shares_reassemble_on_rank0(world, frames, in_flight, shares):
    # the link-aware deal -- every rank its own share of the tiles (a rank may have none), each shard's
    # own bytes point to point, the destination's own shard scattered from where it was traced; "auto" runs the
    # calibration (shard loop, frame loop, one exchange; the destination broadcasts the shares) over gloo
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    mp.spawn(_worker, args=(world, _free_port(), frames, q, in_flight, shares), nprocs=world, join=True)
    assert q.get(timeout=30) == 0.0


def test_partitions_cover_every_tile_once_and_follow_the_shares():
    import numpy as np
    from rtgs.parallel import Partition, balanced_share
    for total in (9, 8160, 32400, 1):
        for shares in ([1], [1, 1], [0.77, 0.23], [0.3] + [0.1] * 7, [5, 0, 1], [0, 0, 1]):
            p = Partition.weighted(total, shares)
            assert sum(p.counts) == total and p.world == len(shares) and not p.arithmetic
            owner, slot = p.tile_map >> 24, p.tile_map & 0xFFFFFF
            for r in range(p.world):
                assert np.array_equal(p.tiles[r], np.nonzero(owner == r)[0])          # slots in increasing tile order
                assert np.array_equal(slot[p.tiles[r]], np.arange(p.counts[r]))
            if total >= 8160:
                got, want = np.array(p.shares()), np.array(shares, float) / sum(shares)
                assert np.abs(got - want).max() < 1.0 / Partition.PERIOD
        rr = Partition.round_robin(total, 3)
        assert rr.arithmetic and [c for c in rr.counts] == [len(range(r, total, 3)) for r in range(3)]
    with pytest.raises(ValueError):
        Partition([np.array([0, 1]), np.array([1, 2])], 3, "overlap")
    with pytest.raises(ValueError):
        Partition.weighted(9, [0, 0])
    # the share the destination keeps: link-bound -> most of the frame, link free -> the equal deal
    assert balanced_share(1, 1.0, 1.0, 1.0) == 1.0
    assert balanced_share(8, 37e-6, 150e-6, 1e-9) == 1 / 8
    x2 = balanced_share(2, 80e-6, 150e-6, 200e-6)
    assert 0.5 < x2 < 1.0
    # at the balance point the destination's trace time equals one peer's transfer time
    c = (150e-6 - 80e-6) / 0.5; p0 = 80e-6 - c / 2
    assert abs((p0 + c * x2) - 200e-6 * 2 * (1 - x2)) < 1e-9


def test_partition_properties_hypothesis():
    # any shares, any frame: a partition (every tile once), shares honoured to a residue, the map inverts the lists;
    # the balanced share grows with the exchange time and never leaves [1/N, 1]
    import numpy as np
    from hypothesis import given, settings, strategies as st
    from rtgs.parallel import Partition, balanced_share

    @settings(max_examples=60, deadline=None)
    @given(st.integers(1, 40_000), st.lists(st.floats(0.0, 10.0), min_size=1, max_size=9).filter(lambda v: sum(v) > 1e-3))
    def check(total, shares):
        p = Partition.weighted(total, shares)
        assert sum(p.counts) == total and len(p.tiles) == len(shares)
        owner = p.tile_map >> 24
        for r, t in enumerate(p.tiles):
            assert np.all(owner[t] == r) and np.all(np.diff(t) > 0)
            assert np.array_equal(p.tile_map[t] & 0xFFFFFF, np.arange(t.size))
        want = np.array(shares) / sum(shares)
        if total >= 64 * 50:
            assert np.abs(np.array(p.shares()) - want).max() <= 1.0 / Partition.PERIOD + 64.0 / total

    check()

    @settings(max_examples=60, deadline=None)
    @given(st.integers(2, 16), st.floats(1e-6, 1e-3), st.floats(1.0, 8.0), st.floats(0.0, 1e-2), st.floats(0.0, 1e-2))
    def share(n, t_shard, ratio, tx1, tx2):
        t_frame = t_shard * ratio
        lo, hi = sorted((tx1, tx2))
        a, b = balanced_share(n, t_shard, t_frame, lo), balanced_share(n, t_shard, t_frame, hi)
        assert 1.0 / n - 1e-12 <= a <= b + 1e-12 <= 1.0 + 1e-12

    share()

    for (w, h, world) in [(1920, 1080, 8), (3840, 2160, 8), (800, 600, 3), (33, 17, 2)]:
        p = Partition.wedges(w, h, 16, world)
        assert p.compact and sum(p.counts) == p.total and max(p.counts) - min(p.counts) <= 1
