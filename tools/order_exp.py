"""Experiment: how much does a MEASURED per-tile cost order gain over the centre-distance prior?
pass 1 (RTGS_HIP_LIB = a -DRTGS_STAMPS=1 build):  python tools/order_exp.py measure
pass 2 (production build):                         python tools/order_exp.py time"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "rt-gaussian-splat-renderer_amd"), ROOT]
import numpy as np, torch
import rtgs
from rtgs import synthetic, _native as nat
from rtgs.parallel import HipBackend, cost_sorted_order
cfg = synthetic.CONFIGS[os.environ.get("RTGS_CFG", "c3")]
arrays = synthetic.make_config_scene(cfg); cam = synthetic.bench_camera(cfg.width, cfg.height)
scene = rtgs.Scene.from_arrays(**arrays)
camera = rtgs.Camera(cam["position"], cam["rotation"], (cfg.width, cfg.height), (cam["fx"], cam["fy"]))
b = HipBackend(scene, camera, cfg.depth); T = b.tile
total = ((cfg.width + T - 1) // T) * ((cfg.height + T - 1) // T)
b.project()
OUT = os.path.join(ROOT, "gpurun_out", "order_exp"); os.makedirs(OUT, exist_ok=True)
WORLDS = (8, 4, 2)
if sys.argv[1] == "measure":
    for world in WORLDS:
        per = (total + world - 1) // world
        compact = b.alloc(per, T, T, 3)
        stamps = torch.zeros((per, 8), dtype=torch.int64, device="cuda")
        params = nat.RtgsTraceParams(cfg.depth, 0.0, 0, world, nat.TRACE_COMPACT)
        for _ in range(3):
            nat.check(nat.lib().rtgs_trace(scene.native(), camera.native(), params, nat.ptr(compact), stamps.data_ptr(),
                                           None, 1, nat.ptr(b.workspace), b.workspace.numel(), 1, nat.stream()))
        torch.cuda.synchronize()
        raw = stamps.cpu().numpy()
        np.save(os.path.join(OUT, f"stamps_w{world}.npy"), raw)
        print("measured world", world, "blocks", per)
else:
    def snake(desc, groups=256):
        # desc: tiles by cost, heaviest first.  Block b, b+groups, b+2*groups ... share a CU: deal
        # the ranks boustrophedon so that every CU's sum is about the same.
        n = len(desc); rows = (n + groups - 1) // groups
        out = np.full(rows * groups, -1, np.int64)
        for r in range(rows):
            seg = desc[r * groups:(r + 1) * groups]
            idx = np.arange(len(seg)) if r % 2 == 0 else groups - 1 - np.arange(len(seg))
            out[r * groups + idx] = seg
        return out[out >= 0]
    for world in WORLDS:
        per = (total + world - 1) // world
        raw = np.load(os.path.join(OUT, f"stamps_w{world}.npy"))
        life = (raw[:, 4] - raw[:, 0]).astype(np.float64)
        ins = ((raw[:, 5] >> 32) & 0xFFFF).astype(np.float64)
        compact = b.alloc(per, T, T, 3)
        orders = {
            "none": None,
            "centre prior": cost_sorted_order(cfg.width, cfg.height, T, 0, world, "centre"),
            "xcd wedges": cost_sorted_order(cfg.width, cfg.height, T, 0, world, "wedge"),
            "measured desc": np.argsort(-life, kind="stable"),
            "measured snake": snake(np.argsort(-life, kind="stable")),
            "insertions desc": np.argsort(-ins, kind="stable"),
            "insertions snake": snake(np.argsort(-ins, kind="stable")),
            "measured asc": np.argsort(life, kind="stable"),
        }
        res = []
        for name, o in orders.items():
            od = None if o is None else torch.from_numpy(np.ascontiguousarray(o, dtype=np.int32)).cuda()
            if od is not None: assert sorted(o.tolist()) == list(range(per)), name
            params = nat.RtgsTraceParams(cfg.depth, 0.0, 0, world, nat.TRACE_COMPACT, None if od is None else od.data_ptr())
            def go():
                nat.check(nat.lib().rtgs_trace(scene.native(), camera.native(), params, nat.ptr(compact), None,
                                               None, 1, nat.ptr(b.workspace), b.workspace.numel(), 1, nat.stream()))
            for _ in range(5): go()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(50): go()
            e1.record(); torch.cuda.synchronize()
            res.append(f"{name} {e0.elapsed_time(e1)/50*1e3:.1f}")
        print(f"1/{world}: " + " | ".join(res) + " us", flush=True)
