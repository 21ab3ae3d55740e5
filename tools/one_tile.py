"""Diagnostic: ONE workgroup alone on the GPU (a heavy c3 tile): the latency chain of a tile without any
contention (needs a -DRTGS_STAMPS=1 build in RTGS_HIP_LIB)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "rt-gaussian-splat-renderer_amd"), ROOT]
import numpy as np, torch
import rtgs
from rtgs import synthetic, _native as nat
from rtgs.parallel import HipBackend
cfg = synthetic.CONFIGS[os.environ.get("RTGS_CFG", "c3")]
arrays = synthetic.make_config_scene(cfg); cam = synthetic.bench_camera(cfg.width, cfg.height)
scene = rtgs.Scene.from_arrays(**arrays)
camera = rtgs.Camera(cam["position"], cam["rotation"], (cfg.width, cfg.height), (cam["fx"], cam["fy"]))
b = HipBackend(scene, camera, 16); T = b.tile
tx = (cfg.width + T - 1) // T
total = tx * ((cfg.height + T - 1) // T)
b.project()
for (ti, tj) in [(60, 33), (46, 44), (10, 5), (60, 60)]:
    tile = tj * tx + ti
    for split in (0,):
        compact = b.alloc(1, T, T, 3)
        stamps = torch.zeros((1, 14), dtype=torch.int64, device="cuda")
        params = nat.RtgsTraceParams(cfg.depth, 0.0, tile, total, nat.TRACE_COMPACT)
        for _ in range(3):
            nat.check(nat.lib().rtgs_trace(scene.native(), camera.native(), params, nat.ptr(compact), stamps.data_ptr(),
                                           None, 1, nat.ptr(b.workspace), b.workspace.numel(), 1, nat.stream()))
        torch.cuda.synchronize()
        raw = stamps.cpu().numpy()[0]
        d = np.diff(raw[:5])
        packed = int(raw[5]); it, hit, ins = packed & 0xFFFF, (packed >> 16) & 0xFFFF, (packed >> 32) & 0xFFFF
        clk = (raw[4] - raw[0]) / max(raw[9] - raw[8], 1) * 100
        print(f"tile ({ti},{tj}) split={split}: scan {d[0]} expand {d[1]} process {d[2]} composite {d[3]} cycles (total {raw[4]-raw[0]}, "
              f"{(raw[4]-raw[0])/clk:.1f} us at {clk:.0f} MHz); wave 0: iterations {it}, hits {hit}, insertions {ins}; "
              f"front {raw[6]/max(ins,1):.0f} + network {raw[7]/max(ins,1):.0f} cycles per insertion; "
              f"rounds {raw[13]}: expansion {raw[10]}, end-of-round barrier waits {raw[11]}, merge {raw[12]} cycles")
