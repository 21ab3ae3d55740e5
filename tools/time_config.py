import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "rt-gaussian-splat-renderer_amd"), ROOT]
import torch, rtgs
from rtgs import synthetic
name = sys.argv[1]
cfg = synthetic.CONFIGS[name]
arrays = synthetic.make_config_scene(cfg); cam = synthetic.bench_camera(cfg.width, cfg.height)
scene = rtgs.Scene.from_arrays(**arrays)
camera = rtgs.Camera(cam["position"], cam["rotation"], (cfg.width, cfg.height), (cam["fx"], cam["fy"]))
tr = rtgs.RayTracer((cfg.width, cfg.height), scene, camera)
for _ in range(3): tr.trace(cfg.depth)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(20): tr.trace(cfg.depth)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"{name}: {ms:.3f} ms/frame = {cfg.rays/ms/1e3:.0f} Mrays/s")
