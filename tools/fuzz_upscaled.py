"""One tests/scenes.py::fuzz_case scene re-rendered at k x the resolution (same field of view):
sampled pixels vs the C oracle.  usage: fuzz_upscaled.py SEED [k ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "rt-gaussian-splat-renderer_amd"), ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import rtgs, scenes
from oracle import c_oracle
seed = int(sys.argv[1])
arrays, cam, depth = scenes.fuzz_case(seed)[:3]
for k in [int(v) for v in sys.argv[2:]] or [1, 4, 8, 12]:
    w, h = cam["width"] * k, cam["height"] * k
    c = dict(cam, width=w, height=h, fx=float(np.float32(cam["fx"] * k)), fy=float(np.float32(cam["fy"] * k)))
    scene = rtgs.Scene.from_arrays(**arrays)
    camera = rtgs.Camera(c["position"], c["rotation"], (w, h), (c["fx"], c["fy"]))
    tr = rtgs.RayTracer((w, h), scene, camera); tr.trace(depth); torch.cuda.synchronize()
    rng = np.random.default_rng(seed)
    pix = np.stack([rng.integers(0, w, 8000), rng.integers(0, h, 8000)], -1).astype(np.int32)
    want = c_oracle.render(arrays, c, depth, pix, eps_q=1e-4 if arrays["scale"].max() > 1 else 1e-5)
    got = tr.sample_buf.cpu().numpy()[pix[:, 0], pix[:, 1]]
    err = np.abs(got - want["sample"]).max(axis=1); solid = ~want["fragile"]
    i = int(np.argmax(np.where(solid, err, 0)))
    print(f"seed {seed} x{k} {w}x{h}: max err {err[solid].max():.2e} at {pix[i].tolist()}, >3e-5: {int((err[solid] > 3e-5).sum())}, fragile {int((~solid).sum())}", flush=True)
