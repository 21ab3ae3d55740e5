"""Capture-like scenes with a growing share of huge background Gaussians: ms/frame by slab ordering."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "rt-gaussian-splat-renderer_amd"), ROOT]
import numpy as np, torch
import rtgs
from rtgs import synthetic, scene as scene_mod
from rtgs.parallel import HipBackend
cfg = synthetic.CONFIGS["cap"]
cam = synthetic.bench_camera(cfg.width, cfg.height)
for huge_share, huge_mult in [(0.0, 0), (0.002, 40.0), (0.01, 40.0), (0.002, 150.0)]:
    arrays = synthetic.make_config_scene(cfg)
    n = arrays["position"].shape[0]
    rng = np.random.default_rng(5)
    k = int(n * huge_share)
    if k:
        pick = rng.permutation(n)[:k]
        s0 = (8.727 * cfg.aspect / n) ** 0.5
        arrays["scale"][pick] = np.float32(huge_mult * s0) * np.exp2(rng.random((k, 3)) - 0.5).astype(np.float32)
        arrays["opacity"][pick] = np.float32(0.05)
    out = []
    for mode in (False, True):
        scene_mod.HUGE_LAST = mode
        scene = rtgs.Scene.from_arrays(**arrays)
        camera = rtgs.Camera(cam["position"], cam["rotation"], (cfg.width, cfg.height), (cam["fx"], cam["fy"]))
        b = HipBackend(scene, camera, 16)
        fb = b.alloc(cfg.width, cfg.height, 3)
        for _ in range(5): b.trace(fb, 0, 1, False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): b.trace(fb, 0, 1, False)
        torch.cuda.synchronize()
        out.append("huge last %s: %.3f ms" % (mode, (time.perf_counter() - t0) / 30 * 1e3))
    print(f"huge share {huge_share} x{huge_mult}: " + ", ".join(out), flush=True)
