"""Counting build with RTGS_SH_STAGE=2: how many composited entries were shaded from the LDS staging vs from memory.
usage: RTGS_COUNT_LIB=path python tools/stage_count.py c3sh"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "rt-gaussian-splat-renderer_amd"), ROOT]
import torch, rtgs
from rtgs import synthetic, _native as nat
cfg = synthetic.CONFIGS[sys.argv[1]]
arrays = synthetic.make_config_scene(cfg); cam = synthetic.bench_camera(cfg.width, cfg.height)
scene = rtgs.Scene.from_arrays(**arrays)
camera = rtgs.Camera(cam["position"], cam["rotation"], (cfg.width, cfg.height), (cam["fx"], cam["fy"]))
cl = nat.load_library(os.environ["RTGS_COUNT_LIB"], {"rtgs_debug_counters": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int32])})
out = torch.zeros((cfg.width, cfg.height, 3), device="cuda")
ws = torch.empty(cl.rtgs_workspace_bytes(scene.num_gaussians), dtype=torch.uint8, device="cuda")
params = nat.RtgsTraceParams(cfg.depth, 0.0, 0, 1, 0, None)
buf = (ctypes.c_ulonglong * 8)()
cl.rtgs_debug_counters(None, 1)
nat.check(cl.rtgs_trace(scene.native(), camera.native(), params, nat.ptr(out), None, None, 1, nat.ptr(ws), ws.numel(), 0, nat.stream()))
cl.rtgs_debug_counters(ctypes.addressof(buf), 1)
c = [int(v) for v in buf]
print(cfg.name, "waves", c[7], "insertion runs", c[2], "staged entries", c[5], "gathered entries", c[6], "mean", out.mean().item())
