"""Time k_trace with one and two waves per 8x8 tile over launch sizes (c3 shards of rank 0)."""
import os, sys
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
    b = HipBackend(scene, camera, cfg.depth)
    T = b.tile
    total = ((cfg.width + T - 1) // T) * ((cfg.height + T - 1) // T)
    b.project()
    for world in ((32, 16, 12, 10, 8, 7, 6, 5, 4) if name == "c3" else (8, 4, 2, 1)):
        per = (total + world - 1) // world
        compact = b.alloc(per, T, T, 3)
        res = []
        for split in (False, True):
            for _ in range(5): b.trace(compact, 0, world, compact=True, projected=True, split=split)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(50): b.trace(compact, 0, world, compact=True, projected=True, split=split)
            e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 50 * 1e3)
        print(f"{name} 1/{world}: {per} tiles: one wave {res[0]:.1f} us, two waves {res[1]:.1f} us", flush=True)
