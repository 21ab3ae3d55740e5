"""Transmittance cut-off on an opaque scene: ms/frame without it, with it (compositing + traversal cut), and --
with RTGS_HIP_LIB pointing at a -DRTGS_CUT_TRAVERSAL=0 build -- with the compositing cut alone."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "rt-gaussian-splat-renderer_amd"), ROOT]
import numpy as np, torch
import rtgs
from rtgs import synthetic
from rtgs.parallel import HipBackend
cfg = synthetic.CONFIGS["c3"]
cam = synthetic.bench_camera(cfg.width, cfg.height)
for name, scale_mul, opa, huge in (("c3 as is", 1.0, None, 0), ("c3 scales x2, opacity 0.8-0.98", 2.0, (0.8, 0.18), 0),
                                   ("c3 opacity 0.8-0.98 + 1 % huge (40 x s0) background", 1.0, (0.8, 0.18), 0.01),
                                   ("c3 scales x2, opacity 0.9-0.99 + 3 % huge", 2.0, (0.9, 0.09), 0.03)):
    arrays = synthetic.make_config_scene(cfg)
    arrays["scale"] = (arrays["scale"] * np.float32(scale_mul)).astype(np.float32)
    if opa:
        arrays["opacity"] = (opa[0] + opa[1] * (arrays["opacity"] - 0.1) / 0.8).astype(np.float32)
    if huge:
        n = arrays["position"].shape[0]
        rng = np.random.default_rng(5)
        pick = rng.permutation(n)[:int(n * huge)]
        s0 = (8.727 * cfg.aspect / n) ** 0.5
        arrays["scale"][pick] = np.float32(40 * s0) * np.exp2(rng.random((len(pick), 3)) - 0.5).astype(np.float32)
        arrays["position"][pick, 2] = -1.0 + 0.5 * rng.random(len(pick)).astype(np.float32)
    scene = rtgs.Scene.from_arrays(**arrays)
    camera = rtgs.Camera(cam["position"], cam["rotation"], (cfg.width, cfg.height), (cam["fx"], cam["fy"]))
    out = []
    for cut in (0.0, 1e-2, 1e-3):
        b = HipBackend(scene, camera, 16, t_cutoff=cut)
        fb = b.alloc(cfg.width, cfg.height, 3)
        b.project()
        for _ in range(5): b.trace(fb, 0, 1, False, projected=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(30): b.trace(fb, 0, 1, False, projected=True)
        e1.record(); torch.cuda.synchronize()
        out.append("t_cutoff %g: %.3f ms" % (cut, e0.elapsed_time(e1) / 30))
    print(f"{name}: k_trace " + ", ".join(out), flush=True)
