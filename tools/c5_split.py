import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "rt-gaussian-splat-renderer_amd"), ROOT]
import torch, rtgs
from rtgs import synthetic
cfg = synthetic.CONFIGS["c5"]
arrays = synthetic.make_config_scene(cfg); cam = synthetic.bench_camera(cfg.width, cfg.height)
for label, sh in (("with SH", arrays["sh"]), ("no SH", None)):
    a = dict(arrays); a["sh"] = sh
    scene = rtgs.Scene.from_arrays(**a)
    camera = rtgs.Camera(cam["position"], cam["rotation"], (cfg.width, cfg.height), (cam["fx"], cam["fy"]))
    tr = rtgs.RayTracer((cfg.width, cfg.height), scene, camera)
    for _ in range(3): tr.trace(cfg.depth)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): tr.trace(cfg.depth)
    e1.record(); torch.cuda.synchronize()
    print(f"c5 {label}: {e0.elapsed_time(e1)/10:.3f} ms/frame")
