import os, sys, time, cProfile, pstats
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [os.path.join(ROOT, "rt-gaussian-splat-renderer_amd"), ROOT]
import torch, rtgs
from rtgs import synthetic
arrays = synthetic.make_scene(100_000, 16 / 9, 1003)
cam = synthetic.bench_camera(1920, 1080)
scene = rtgs.Scene.from_arrays(**arrays)
camera = rtgs.Camera(cam["position"], cam["rotation"], (1920, 1080), (cam["fx"], cam["fy"]))
tr = rtgs.RayTracer((1920, 1080), scene, camera)
for _ in range(32): tr.sample(16)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(160): tr.sample(16)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host issue per call %.1f us, total per call %.1f us" % ((t1 - t0) / 160 * 1e6, (t2 - t0) / 160 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(160): tr.sample(16)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
