"""EVERY rank's tile-parallel frame loop on ONE GPU, one after the other (VERDICT r3 #2c): for N = 2, 4, 8 and the
equal and the link-aware deals, each rank's loop (project + trace on F render streams, event chain, the destination's
scatter; the exchange itself stubbed out) in microseconds per frame, the maximum over the ranks, and -- labelled as a
PROJECTION -- the frame time with one peer's bytes crossing an xGMI link at an assumed rate.  The shares of the
link-aware deal come from rtgs.parallel.balanced_share with the exchange time that rate implies (on hardware the
renderer measures it: TileParallelRenderer(partition="auto")).

usage: python tools/rank_loops.py [cfg ...]    env: WORLDS="2 4 8"  LINK_GBS="50 64 77"  F=2  DEALS=wedges (also the sector deal)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "rt-gaussian-splat-renderer_amd"), ROOT]
import torch
import rtgs
from rtgs import synthetic
from rtgs import _native as nat
from rtgs.parallel import Partition, TileParallelRenderer, _no_gather, balanced_share, tiles_of

torch.cuda.set_device(0)
F = int(os.environ.get("F", "2"))
worlds = [int(v) for v in os.environ.get("WORLDS", "2 4 8").split()]
links = [float(v) for v in os.environ.get("LINK_GBS", "50 64 77").split()]


def loop_us(r, n):
    for _ in range(20):
        r.render_frame()
    r.finish(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r.render_frame()
    r.finish(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for name in (sys.argv[1:] or ["c3", "c5"]):
    cfg = synthetic.CONFIGS[name]
    arrays = synthetic.make_config_scene(cfg); cam = synthetic.bench_camera(cfg.width, cfg.height)
    scene = rtgs.Scene.from_arrays(**arrays)
    camera = rtgs.Camera(cam["position"], cam["rotation"], (cfg.width, cfg.height), (cam["fx"], cam["fy"]))
    n = 300 if name != "c5" else 100
    total = tiles_of(cfg.width, cfg.height, nat.tile_size())
    frame_bytes = cfg.width * cfg.height * 12
    one = TileParallelRenderer(scene, camera, depth=cfg.depth, rank=0, world=1)
    t_frame = loop_us(one, n)
    print(f"== {name}: single-GPU frame loop {t_frame:.1f} us ({cfg.width}x{cfg.height}, {frame_bytes / 1e6:.1f} MB of pixels), {F} frames in flight per rank")
    for world in worlds:
        eq = TileParallelRenderer(scene, camera, depth=cfg.depth, rank=0, world=world, gather_fn=_no_gather, frames_in_flight=F)
        t_shard = loop_us(eq, n); eq.close()
        deals = [("equal", None, None)]
        if "wedges" in os.environ.get("DEALS", ""):
            deals.append(("wedges (one compact sector per rank)", "wedges", None))
        for gbs in links:
            t_x = frame_bytes / world / (gbs * 1e3)           # us for one 1/N shard over one link
            x = balanced_share(world, t_shard, t_frame, t_x)
            if x > 1.02 / world:
                sh = [(1 - x) / (world - 1)] * world; sh[0] = x
                deals.append((f"weighted for {gbs:.0f} GB/s (x = {x:.2f})", sh, gbs))
        for label, shares, gbs in deals:
            part = (Partition.round_robin(total, world) if shares is None else
                    Partition.wedges(cfg.width, cfg.height, nat.tile_size(), world) if isinstance(shares, str) else
                    Partition.weighted(total, shares))
            per = []
            for rank in range(world):
                r = TileParallelRenderer(scene, camera, depth=cfg.depth, rank=rank, world=world, gather_fn=_no_gather,
                                         p2p_fn=lambda mine, landing: [], partition=part, frames_in_flight=F)
                per.append(loop_us(r, n)); r.close()
            worst = max(per)
            peer_bytes = max(part.counts[1:]) * 256 * 12
            proj = " | ".join(f"{g:.0f} GB/s: link {peer_bytes / (g * 1e3):.0f} us -> frame {max(worst, peer_bytes / (g * 1e3)):.0f} us = {t_frame / max(worst, peer_bytes / (g * 1e3)):.2f}x"
                             for g in ([gbs] if gbs else links))
            print(f"{name} N={world} {label}: deal {part.name}; rank loops us " + " ".join(f"{v:.0f}" for v in per) +
                  f"; max {worst:.0f} (rank {per.index(worst)}); PROJECTION {proj}")
    del scene
