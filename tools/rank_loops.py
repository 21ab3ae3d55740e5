"""EVERY rank's tile-parallel frame loop on ONE GPU, one after the other (VERDICT r3 #2c, r4 #1): for N = 2, 4, 8, for
the single root (every frame lands on rank 0: one gather per batch) and for ROTATING roots (frame f lands on rank
f mod N: one all_to_all per batch), each rank's loop -- rtgs_shard_frames (project + trace of B frames over F render
streams), the event chain, the landing scatter of the frames that land on that rank; the exchange itself stubbed out --
in microseconds per frame, the maximum over the ranks, and -- labelled as a PROJECTION -- the frame time with the
bytes a link has to carry per frame at an assumed rate (root: one 1/N shard per frame into rank 0 over each link;
rotate: one 1/N shard per N frames over each link).

usage: python tools/rank_loops.py [cfg ...]   env: WORLDS="2 4 8"  LINK_GBS="50 64 77"  F="2 3"  B="1 8"  DEALS="equal wedges weighted"
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "rt-gaussian-splat-renderer_amd"), ROOT]
import torch
import rtgs
from rtgs import synthetic
from rtgs import _native as nat
from rtgs.parallel import Partition, TileParallelRenderer, _no_all_to_all, _no_gather, balanced_share, tiles_of

torch.cuda.set_device(0)
Fs = [int(v) for v in os.environ.get("F", "2 3").split()]
Bs = [int(v) for v in os.environ.get("B", "1 8").split()]
worlds = [int(v) for v in os.environ.get("WORLDS", "2 4 8").split()]
links = [float(v) for v in os.environ.get("LINK_GBS", "50 64 77").split()]
deals = os.environ.get("DEALS", "equal").split()


def loop_us(r, n):
    r.render_frames(count=24)
    r.finish(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    r.render_frames(count=n)
    r.finish(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def host_us(r, n):
    """Host time of queueing n frames (the GPU drains afterwards, untimed)."""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r.render_frames(count=n)
    dt = time.perf_counter() - t0
    r.finish(); torch.cuda.synchronize()
    return dt / n * 1e6


for name in (sys.argv[1:] or ["c3", "c5"]):
    cfg = synthetic.CONFIGS[name]
    arrays = synthetic.make_config_scene(cfg); cam = synthetic.bench_camera(cfg.width, cfg.height)
    scene = rtgs.Scene.from_arrays(**arrays)
    camera = rtgs.Camera(cam["position"], cam["rotation"], (cfg.width, cfg.height), (cam["fx"], cam["fy"]))
    n = 480 if name != "c5" else 96
    total = tiles_of(cfg.width, cfg.height, nat.tile_size())
    frame_bytes = cfg.width * cfg.height * 12
    one = TileParallelRenderer(scene, camera, depth=cfg.depth, rank=0, world=1)
    t_frame = loop_us(one, n)
    print(f"== {name}: single-GPU frame loop {t_frame:.1f} us ({cfg.width}x{cfg.height}, {frame_bytes / 1e6:.1f} MB of pixels)", flush=True)
    for world in worlds:
        parts = []
        if "equal" in deals:
            parts.append(("equal", Partition.round_robin(total, world)))
        if "wedges" in deals:
            parts.append(("wedges", Partition.wedges(cfg.width, cfg.height, nat.tile_size(), world)))
        if "all0" in deals:       # rank 0 renders EVERY tile through an explicit list (a whole frame through the shard path)
            parts.append(("all0", Partition.weighted(total, [1.0] + [0.0] * (world - 1))))
        for dname, part in parts:
            for landing in ("root", "rotate"):
                for B in Bs:
                    if landing == "rotate" and B < world:
                        continue
                    for F in Fs:
                        per, host = [], []
                        for rank in ([int(v) for v in os.environ["RANKS"].split()] if os.environ.get("RANKS") else range(world)):
                            r = TileParallelRenderer(scene, camera, depth=cfg.depth, rank=rank, world=world, gather_fn=_no_gather,
                                                     p2p_fn=lambda mine, landing: [], a2a_fn=_no_all_to_all, partition=part,
                                                     frames_in_flight=F, gather_batch=B, landing=landing)
                            per.append(loop_us(r, n)); host.append(host_us(r, n))
                            if os.environ.get("DEBUG_PTRS") and r._comm:
                                cm = r._comm
                                print(f"   rank {rank}: loop {per[-1]:.1f} us  ws " + " ".join(hex(w.data_ptr()) for w in cm["ws"]) + "  compact " +
                                      " ".join(hex(c.data_ptr()) for c in r.compact) + "  gathered " + " ".join(hex(g.data_ptr()) for g in r.gathered) +
                                      "  fbs " + " ".join(hex(f.data_ptr()) for f in r._fbs) + f"  ws bytes {cm['ws'][0].numel()}", flush=True)
                            r.close()
                            Beff = r.gather_batch
                        worst = max(per)
                        shard = max(part.counts) * 256 * 12
                        link_bytes = shard if landing == "root" else shard / world       # per frame over one link
                        proj = " | ".join(f"{g:.0f} GB/s: link {link_bytes / (g * 1e3):.0f} us -> frame {max(worst, link_bytes / (g * 1e3)):.0f} us = "
                                          f"{t_frame / max(worst, link_bytes / (g * 1e3)):.2f}x" for g in links)
                        print(f"{name} N={world} {dname} landing={landing} B={Beff} F={F}: rank loops us " + " ".join(f"{v:.0f}" for v in per) +
                              f"; max {worst:.0f} (rank {per.index(worst)}) = {t_frame / worst:.2f}x without the link; host us/frame " +
                              " ".join(f"{v:.0f}" for v in host) + f"; PROJECTION {proj}", flush=True)
    del scene
