"""Where a moving-camera viewer sample goes (RayTracer.sample() x depth with a new pose per sample): projection, the
traversal that writes the peel lists, the sixteen listed peels.  HIP events, ms.   usage: python tools/viewer_breakdown.py [cfg ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "rt-gaussian-splat-renderer_amd"), ROOT]
import numpy as np, torch
import rtgs
from rtgs import synthetic, _native as nat
from rtgs.utils.types import vec3

def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize()
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for name in (sys.argv[1:] or ["c3", "c3sh"]):
    cfg = synthetic.CONFIGS[name]
    arrays = synthetic.make_config_scene(cfg); cam = synthetic.bench_camera(cfg.width, cfg.height)
    scene = rtgs.Scene.from_arrays(**arrays)
    camera = rtgs.Camera(cam["position"], cam["rotation"], (cfg.width, cfg.height), (cam["fx"], cam["fy"]))
    tr = rtgs.RayTracer((cfg.width, cfg.height), scene, camera)
    lib = nat.lib()
    ws = tr._ws()
    turn = [0]
    def pose():
        turn[0] += 1
        return (0.2 * float(np.sin(0.05 * turn[0])), 0.1 * float(np.cos(0.05 * turn[0])), 3.0), tuple(cam["rotation"])
    for _ in range(16): tr.sample(16)          # allocates the lists
    def project():
        p, r = pose()
        nat.check(lib.rtgs_project(scene.native(), camera.native(p, r), nat.ptr(ws), ws.numel(), nat.stream()), "project")
    def lists(projected):
        p, r = pose()
        nat.check(lib.rtgs_trace_lists(scene.native(), camera.native(p, r), nat.ptr(tr._tile_order(0, 1)), nat.ptr(tr._lists),
                                       tr._lists.numel(), nat.ptr(ws), ws.numel(), projected, nat.stream()), "lists")
    def trace():
        p, r = pose()
        params = nat.RtgsTraceParams(16, 0.0, 0, 1, 0, nat.ptr(tr._tile_order(0, 1)))
        nat.check(lib.rtgs_trace(scene.native(), camera.native(p, r), params, nat.ptr(tr.sample_buf), None, None, 1, nat.ptr(ws),
                                 ws.numel(), 1, nat.stream()), "trace")
    def peels():
        for k in range(16):
            nat.check(lib.rtgs_peel(scene.native(), camera.native(), nat.ptr(tr._lists), tr._lists.numel(), k, 1 if k == 0 else 0,
                                    nat.ptr(tr.sample_buf), nat.ptr(tr._attenuation), nat.stream()), "peel")
    t_proj = timed(project, 50)
    t_lists_all = timed(lambda: lists(0), 50)
    t_lists = timed(lambda: lists(1), 50)
    t_trace = timed(trace, 50)
    t_peels = timed(peels, 20)
    def whole():
        p, r = pose()
        camera.position = vec3(p)
        tr.clear_sample(); tr.num_steps = 0
        for _ in range(16): tr.sample(16)
    t_whole = timed(whole, 20)
    print(f"{name}: project+bin {t_proj:.4f}  traversal-to-lists alone {t_lists:.4f} (with its projection {t_lists_all:.4f}; the compositing "
          f"k_trace of the same pose {t_trace:.4f})  16 listed peels {t_peels:.4f} ({t_peels / 16 * 1e3:.1f} us each)  "
          f"whole moving-camera sample through RayTracer {t_whole:.4f} ms", flush=True)
