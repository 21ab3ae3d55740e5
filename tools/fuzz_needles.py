"""GPU fuzz with capture-like Gaussians: log-uniform scales, anisotropy up to ~300:1."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "rt-gaussian-splat-renderer_amd"), ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import rtgs, scenes
from rtgs import synthetic
from oracle import c_oracle
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
worst = 0.0; bad = []; hist = []
for seed in range(9000, 9000 + n_cases):
    rng = np.random.default_rng(seed)
    w, h = int(rng.integers(64, 200)), int(rng.integers(48, 160))
    n = int(rng.choice([500, 3000, 12000]))
    arrays = synthetic.make_scene(n, w / h, seed)
    lo, hi = np.log(1e-3), np.log(1e-1)
    base = np.exp(rng.uniform(lo, hi, (n, 1)))
    aniso = np.exp(rng.uniform(0, np.log(rng.choice([10.0, 100.0, 300.0])), (n, 3)))
    arrays["scale"] = (base * aniso / aniso.max(axis=1, keepdims=True) * rng.choice([1.0, 3.0])).astype(np.float32)
    cam = synthetic.bench_camera(w, h)
    cam["position"] = scenes.f32_tuple(rng.normal(size=3) * 0.3 + np.array([0, 0, rng.choice([1.5, 3.0, 6.0])]))
    cam["rotation"] = scenes.quat_axis_angle(rng.normal(size=3), rng.uniform(0, 0.4))
    cam["fx"] = cam["fy"] = float(np.float32(h * rng.uniform(0.8, 2.5)))
    scene = rtgs.Scene.from_arrays(**arrays)
    camera = rtgs.Camera(cam["position"], cam["rotation"], (w, h), (cam["fx"], cam["fy"]))
    tr = rtgs.RayTracer((w, h), scene, camera); tr.trace(16); torch.cuda.synchronize()
    got = tr.sample_buf.cpu().numpy().reshape(-1, 3)
    want = c_oracle.render(arrays, cam, 16)
    err = np.abs(got - want["sample"]).max(axis=1); solid = ~want["fragile"]
    e = float(err[solid].max()) if solid.any() else 0.0
    worst = max(worst, e); hist.append(e)
    if e > 1e-4: bad.append((seed, e, n, float((arrays["scale"].max(1) / arrays["scale"].min(1)).max()), float(want["hits"].mean()), float(1 - solid.mean())))
print(f"{n_cases} needle cases: worst {worst:.3e}; median of per-case max {np.median(hist):.2e}; failures {bad}")
