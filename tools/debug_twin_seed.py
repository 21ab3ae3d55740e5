"""tests/test_gpu_parallel_fuzz.py::twin_paths_case SEED taken apart: after how many peels does the single-peel sample() loop
leave sample_all's image, in how many pixels, and what do the two hold there.  usage: python tools/debug_twin_seed.py SEED"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "rt-gaussian-splat-renderer_amd"), ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import rtgs
import test_gpu_parallel_fuzz as F
from rtgs import _native as nat

seed = int(sys.argv[1])
arrays, cam, depth, world = F.twin_paths_case(seed)
w, h = cam["width"], cam["height"]
print(f"seed {seed}: {w}x{h}, n {arrays['position'].shape[0]}, depth {depth}, sh {arrays['sh'] is not None}")
def tracer(lists=True):
    t = rtgs.RayTracer((w, h), rtgs.Scene.from_arrays(**arrays), rtgs.Camera(cam["position"], cam["rotation"], (w, h), (cam["fx"], cam["fy"])))
    t._use_lists = lists
    return t
d = tracer(False)
c = tracer(True)
for k in range(1, min(depth, 16) + 1):
    d.sample(depth)
    c.sample(depth)
    a = tracer()
    a.sample_all(k)
    torch.cuda.synchronize()
    for name, t in (("single-peel", d), ("listed", c)):
        gap = (a.sample_buf - t.sample_buf).abs().amax(dim=-1)
        bad = gap > 2e-5
        if bad.any():
            i, j = [int(v[0]) for v in torch.nonzero(bad)[:1].T]
            print(f"  after {k} peels: {name} differs from sample_all({k}) in {int(bad.sum())} px, max {float(gap.max()):.2e}; first ({i},{j}): "
                  f"{t.sample_buf[i, j].tolist()} vs {a.sample_buf[i, j].tolist()}; attenuation {float(t.attenuation_buf[i, j]):.6f} vs {float(a.attenuation_buf[i, j]):.6f}; "
                  f"start {float(t.camera.cam_ray_field.start[i, j]):.7f} vs {float(a.camera.cam_ray_field.start[i, j]):.7f}")
            break
    else:
        continue
    break
else:
    print("  no difference up to", min(depth, 16), "peels")

# ---- the failing pixel's entries: the peel lists of the listed tracer (== sample_all's sequence) against the single-peel loop
if 'i' in dir():
    T_ = nat.tile_size()
    tx = (w + T_ - 1) // T_
    tiles = tx * ((h + T_ - 1) // T_)
    tile = (j // T_) * tx + i // T_
    sub = ((j % T_) // 8) * 2 + ((i % T_) // 8)
    thread = sub * 64 + (((i % 8) << 3) | (j % 8))
    raw = c._lists
    cnt = raw[:tiles * 256 * 4].view(torch.int32)
    slots = tiles * 16 * 256
    idxp = raw[tiles * 256 * 4: tiles * 256 * 4 + slots * 4].view(torch.int32)
    tp = raw[tiles * 256 * 4 + 2 * slots * 4: tiles * 256 * 4 + 3 * slots * 4].view(torch.float32)
    nlist = int(cnt[tile * 256 + thread])
    sc = c.scene
    pos = sc.position.cpu().numpy(); col = sc.color_opacity.cpu().numpy()
    print(f"  pixel ({i},{j}): tile {tile}, thread {thread}, listed entries {nlist}")
    d2 = tracer(False)
    prevS, prevT = np.zeros(3), 1.0
    for kk in range(min(nlist, 16)):
        g = int(idxp[(tile * 16 + kk) * 256 + thread]); t = float(tp[(tile * 16 + kk) * 256 + thread])
        twins = [int(q) for q in np.nonzero((pos == pos[g]).all(axis=1))[0]]
        d2.sample(depth)
        torch.cuda.synchronize()
        S = d2.sample_buf[i, j].cpu().numpy().astype(np.float64); Tn = float(d2.attenuation_buf[i, j])
        wa = prevT - Tn
        rgb = (S - prevS) / wa if wa > 0 else np.zeros(3)
        st = float(d2.camera.cam_ray_field.start[i, j])
        who = [q for q in twins if np.abs(col[q, :3] - rgb).max() < 2e-3]
        print(f"    entry {kk}: listed g {g} (slab {g // 64}, lane {g % 64}) t {t:.7f}; same-position Gaussians {twins}; single-peel start {st:.7f}, colour matches {who}")
        prevS, prevT = S, Tn

# ---- the slab keys and the tile's slab list as the last projection of tracer `c` left them in its workspace
if 'tile' in dir():
    n_g = sc.num_gaussians
    nn = (n_g + 63) & ~63
    nsl = nn // 64
    nsup = (nsl + 15) // 16
    ws = c._workspace
    off_keys = nn * 16 + nn * (16 + 8 + 1) * 4 + (nsl + nsup) * 16
    keys = ws[off_keys: off_keys + nsl * 4].view(torch.float32).cpu().numpy()
    off_bins = off_keys + ((nsl + 3) & ~3) * 4
    stride = (min(nsl, 1024) + 3) & ~3
    bin_count = ws[off_bins: off_bins + 4096 * 4].view(torch.int32).cpu().numpy()
    off_list = off_bins + 4096 * 4
    off_tile = off_bins + 4096 * (1 + stride) * 4
    tb = ws[off_tile + tile * 256 * 4: off_tile + (tile + 1) * 256 * 4].view(torch.int32).cpu().numpy()
    shift = 2
    while ((tx + (1 << shift) - 1) >> shift) * ((((h + T_ - 1) // T_) + (1 << shift) - 1) >> shift) > 4096:
        shift += 1
    sx = (tx + (1 << shift) - 1) >> shift
    st = ((tile // tx) >> shift) * sx + ((tile % tx) >> shift)
    bl = ws[off_list + st * stride * 4: off_list + (st + 1) * stride * 4].view(torch.int32).cpu().numpy()
    print(f"  slabs {nsl}, supertile {st}: bin count {bin_count[st]}, tile list count {tb[0]}")
    for name, lst, cntl in (("tile list", tb[1:], tb[0]), ("supertile bin", bl, bin_count[st])):
        if cntl > 0:
            l = list(lst[:cntl])
            ks = keys[l]
            print(f"    {name}: sorted by key: {bool((np.diff(ks) >= 0).all())}; positions of the entries' slabs:",
                  {int(s_): (l.index(s_) if s_ in l else None) for s_ in sorted({int(idxp[(tile * 16 + q) * 256 + thread]) // 64 for q in range(nlist)})})
    print("    keys:", {int(s_): float(keys[s_]) for s_ in sorted({int(idxp[(tile * 16 + q) * 256 + thread]) // 64 for q in range(nlist)})})
