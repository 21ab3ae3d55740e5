"""Stress: N Gaussians (default 3 M) with SH at 3840x2160, sampled pixels vs the C oracle + timing."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "rt-gaussian-splat-renderer_amd"), ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import rtgs
from rtgs import synthetic
from oracle import c_oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
w, h = 3840, 2160
t0 = time.time(); arrays = synthetic.make_scene(n, w / h, 4242, sh=True); cam = synthetic.bench_camera(w, h)
print(f"scene {n} built in {time.time()-t0:.1f}s", flush=True)
scene = rtgs.Scene.from_arrays(**arrays)
camera = rtgs.Camera(cam["position"], cam["rotation"], (w, h), (cam["fx"], cam["fy"]))
tr = rtgs.RayTracer((w, h), scene, camera)
tr.trace(16); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): tr.trace(16)
e1.record(); torch.cuda.synchronize()
print(f"{e0.elapsed_time(e1)/5:.3f} ms/frame (k_trace alone: RayTracer.trace reuses the projection of an unchanged pose)", flush=True)
from rtgs.parallel import TileParallelRenderer
r = TileParallelRenderer(scene, camera, depth=16)
for _ in range(2): r.render_frame()
torch.cuda.synchronize(); e0.record()
for _ in range(5): r.render_frame()
e1.record(); torch.cuda.synchronize()
print(f"{e0.elapsed_time(e1)/5:.3f} ms/frame with the per-frame projection and bins", flush=True)
del r
if os.environ.get("RTGS_STRESS_TIME_ONLY") == "1":
    sys.exit(0)
rng = np.random.default_rng(3)
npx = max(64, min(1200, int(3.6e9 // n)))
pix = np.stack([rng.integers(0, w, npx), rng.integers(0, h, npx)], -1).astype(np.int32)
t0 = time.time(); want = c_oracle.render(arrays, cam, 16, pix); print(f"oracle {time.time()-t0:.1f}s", flush=True)
got = tr.sample_buf.cpu().numpy()[pix[:, 0], pix[:, 1]]
err = np.abs(got - want["sample"]).max(axis=1); solid = ~want["fragile"]
print(f"max err robust {err[solid].max():.2e}; fragile {int((~solid).sum())}; mean hits {want['hits'].mean():.1f}; mem {torch.cuda.max_memory_allocated()/2**30:.2f} GiB")
