"""Time k_trace on 1/8, 1/4, 1/2 shards (rank 0's tiles), projection already done."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "rt-gaussian-splat-renderer_amd"), ROOT]
import torch
import rtgs
from rtgs import synthetic
from rtgs.parallel import HipBackend
cfg = synthetic.CONFIGS["c3"]
arrays = synthetic.make_config_scene(cfg); cam = synthetic.bench_camera(cfg.width, cfg.height)
scene = rtgs.Scene.from_arrays(**arrays)
camera = rtgs.Camera(cam["position"], cam["rotation"], (cfg.width, cfg.height), (cam["fx"], cam["fy"]))
b = HipBackend(scene, camera, 16)
T = b.tile
total = ((cfg.width + T - 1) // T) * ((cfg.height + T - 1) // T)
b.project()
for split in (False,):
    out = []
    for world in (16, 8, 4, 2, 1):
        per = (total + world - 1) // world
        compact = b.alloc(per, T, T, 3)
        for _ in range(5): b.trace(compact, 0, world, compact=True, projected=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(50): b.trace(compact, 0, world, compact=True, projected=True)
        e1.record(); torch.cuda.synchronize()
        out.append("1/%d: %.1f us" % (world, e0.elapsed_time(e1) / 50 * 1e3))
    print("   shard trace (%s):" % "one wave per tile", ", ".join(out))
