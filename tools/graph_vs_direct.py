"""Rank 0's per-frame cost of project + trace for 1/world shards: captured hipGraph replay vs direct launches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "rt-gaussian-splat-renderer_amd"), ROOT]
import torch
import rtgs
from rtgs import synthetic
from rtgs.parallel import HipBackend
for name in sys.argv[1:] or ["c3"]:
    cfg = synthetic.CONFIGS[name]
    arrays = synthetic.make_config_scene(cfg); cam = synthetic.bench_camera(cfg.width, cfg.height)
    scene = rtgs.Scene.from_arrays(**arrays)
    camera = rtgs.Camera(cam["position"], cam["rotation"], (cfg.width, cfg.height), (cam["fx"], cam["fy"]))
    b = HipBackend(scene, camera, 16); T = b.tile
    total = ((cfg.width + T - 1) // T) * ((cfg.height + T - 1) // T)
    for world in (8, 4, 2, 1):
        per = (total + world - 1) // world
        out = b.alloc(per, T, T, 3)
        res = []
        for graph in (True, False):
            b.use_graph = graph
            for _ in range(20): b.render(out, 0, world, True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = 300
            for _ in range(n): b.render(out, 0, world, True)
            th = time.perf_counter() - t0
            torch.cuda.synchronize(); ta = time.perf_counter() - t0
            res.append("%s host %.1f us, frame %.1f us" % ("graph" if graph else "direct", th / n * 1e6, ta / n * 1e6))
        print(f"{name} 1/{world} shard (project + trace): " + "; ".join(res))
