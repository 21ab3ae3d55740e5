"""Do kernels of two HIP streams overlap on this box?  k_trace of two different 1/8 shards of c3, one
stream vs two streams (the C ABI takes the stream as an argument), and k_untile beside a trace."""
import os, sys, time
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
per = (total + 7) // 8
c0, c1 = b.alloc(per, T, T, 3), b.alloc(per, T, T, 3)
g8 = b.alloc(8, per, T, T, 3); img = b.alloc(cfg.width, cfg.height, 3)
b.project()
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()

def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6

def serial():
    b.trace(c0, 0, 8, compact=True, projected=True, stream=s0)
    b.trace(c1, 1, 8, compact=True, projected=True, stream=s0)
def parallel():
    b.trace(c0, 0, 8, compact=True, projected=True, stream=s0)
    b.trace(c1, 1, 8, compact=True, projected=True, stream=s1)
def trace_untile_serial():
    b.trace(c0, 0, 8, compact=True, projected=True, stream=s0)
    b.untile(g8, 8, per, img, stream=s0)
def trace_untile_parallel():
    b.trace(c0, 0, 8, compact=True, projected=True, stream=s0)
    b.untile(g8, 8, per, img, stream=s1)
print("two 1/8-shard traces, one stream:  %.1f us per pair" % timeit(serial))
print("two 1/8-shard traces, two streams: %.1f us per pair" % timeit(parallel))
print("trace + untile, one stream:        %.1f us per pair" % timeit(trace_untile_serial))
print("trace + untile, two streams:       %.1f us per pair" % timeit(trace_untile_parallel))
