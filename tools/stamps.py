"""Diagnostic: per-phase s_memtime profile of k_trace (needs a -DRTGS_STAMPS=1 build in RTGS_HIP_LIB)."""
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
total = ((cfg.width + T - 1) // T) * ((cfg.height + T - 1) // T)
b.project()
for world in (1, 8):
    per = (total + world - 1) // world
    compact = b.alloc(per, T, T, 3)
    stamps = torch.zeros((per, 14), dtype=torch.int64, device="cuda")
    params = nat.RtgsTraceParams(cfg.depth, 0.0, 0, world, nat.TRACE_COMPACT)
    for _ in range(3):
        nat.check(nat.lib().rtgs_trace(scene.native(), camera.native(), params, nat.ptr(compact), stamps.data_ptr(),
                                       None, 1, nat.ptr(b.workspace), b.workspace.numel(), 1, nat.stream()))
    torch.cuda.synchronize()
    raw = stamps.cpu().numpy()
    lv = raw[:, 0] > 0
    clk = (raw[lv, 4] - raw[lv, 0]).sum() / max((raw[lv, 9] - raw[lv, 8]).sum(), 1) * 100.0
    span_us = (raw[lv, 9].max() - raw[lv, 8].min()) / 100.0
    print(f"   shader clock inside the kernel: {clk:.0f} MHz (s_memtime ticks per 100 MHz s_memrealtime tick, summed over workgroups); "
          f"first start to last end {span_us:.1f} us")
    packed = raw[:, 5]
    cnts = np.stack([packed & 0xFFFF, (packed >> 16) & 0xFFFF, (packed >> 32) & 0xFFFF], axis=1).astype(np.float64)
    slabs = ((packed >> 48) & 0xFFFF)[raw[:, 0] > 0]
    print(f"   surviving slabs per tile: mean {slabs.mean():.0f}, p50 {np.median(slabs):.0f}, p95 {np.percentile(slabs, 95):.0f}, "
          f"max {slabs.max()}; tiles above 256: {(slabs > 256).mean():.1%}")
    t_front, t_ins = raw[:, 6].astype(np.float64), raw[:, 7].astype(np.float64)
    live_ = raw[:, 0] > 0
    ni = np.maximum(cnts[:, 2], 1)
    print(f"   inside the candidate loop (wave 0): front (load+test+hit path) {t_front[live_].sum()/cnts[live_,2].sum():.0f} cycles/insertion, "
          f"insertion network {t_ins[live_].sum()/cnts[live_,2].sum():.0f} cycles/insertion")
    hwid, xcc = raw[:, 6] * 0, raw[:, 7] * 0
    cu, sh, se = (hwid >> 8) & 0xF, (hwid >> 12) & 1, (hwid >> 13) & 0x7
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    if False:
        print("   first 40 blocks (xcc,se,sh,cu):", [(int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i])) for i in range(40)])
        import collections
        per_cu = collections.defaultdict(list)
        for i in range(len(cuid)): per_cu[int(cuid[i])].append(i)
        sizes = collections.Counter(len(v) for v in per_cu.values())
        print("   distinct CUs:", len(per_cu), "WGs per CU histogram:", dict(sizes))
        ex = list(per_cu.items())[:6]
        print("   blocks sharing a CU (examples):", [v for _, v in ex])
    s = raw[:, :5].astype(np.float64)
    cnts = cnts[s[:, 0] > 0]
    s = s[s[:, 0] > 0]
    d = np.diff(s, axis=1)                      # scan, expand, process, composite
    t0 = s[:, 0].min()
    print(f"world {world}: {len(s)} blocks; kernel span {(s[:,4].max()-t0)/100:.1f} us (100 MHz ticks?)")
    names = ["scan", "expand", "process", "composite"]
    for k, nme in enumerate(names):
        print(f"   {nme:10s} mean {d[:,k].mean():9.0f}  p50 {np.median(d[:,k]):9.0f}  p95 {np.percentile(d[:,k],95):9.0f} ticks")
    life = s[:, 4] - s[:, 0]
    proc = d[:, 2]
    print(f"   wave 0 per WG: iterations mean {cnts[:,0].mean():.1f} max {cnts[:,0].max():.0f}; passed test {cnts[:,1].mean():.1f}; "
          f"insertions {cnts[:,2].mean():.1f} max {cnts[:,2].max():.0f}; process cycles / insertion (sum/sum) {proc.sum()/max(cnts[:,2].sum(),1):.0f}")
    heavy = proc > np.percentile(proc, 90)
    print(f"   heaviest 10%: iterations {cnts[heavy,0].mean():.0f}, insertions {cnts[heavy,2].mean():.0f}, process {proc[heavy].mean():.0f} cycles "
          f"=> {proc[heavy].sum()/cnts[heavy,2].sum():.0f} cycles/insertion")
    print(f"   lifetime   mean {life.mean():9.0f}  p50 {np.median(life):9.0f}  max {life.max():9.0f}; start spread {(s[:,0].max()-t0):.0f}")
    if world == 1:
        tx = (cfg.width + T - 1) // T
        order = np.argsort(-life)
        print("   slowest tiles (tx,ty,cycles):", [(int(i % tx), int(i // tx), int(life[i])) for i in order[:10]])
        grid = np.zeros(((cfg.height + T - 1) // T, tx)); grid.flat[:len(life)] = life
        # coarse 8x10 heat map of mean lifetime (kcycles)
        H, W = grid.shape
        for r in range(7, -1, -1):
            row = grid[r * H // 8:(r + 1) * H // 8]
            print("   " + " ".join(f"{row[:, c * W // 10:(c + 1) * W // 10].mean() / 1000:5.0f}" for c in range(10)))
