"""Which pixels of a depth-1 frame differ from the oracle, and what the listed peel / deeper frames say there."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "rt-gaussian-splat-renderer_amd"), ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import gpu_util as U, scenes
import test_gpu_random as t
from oracle import c_oracle
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 4005
arrays, cam, depth = t.random_case(seed)
rng = np.random.default_rng(seed)
arrays["rotation"] = (arrays["rotation"] * rng.uniform(0.7, 1.4, (arrays["rotation"].shape[0], 1))).astype(np.float32)
cam["rotation"] = scenes.f32_tuple(np.asarray(cam["rotation"]) * rng.uniform(0.8, 1.25))
W, H = cam["width"], cam["height"]
FLAGS = int(os.environ.get("FLAGS", "0"))
for d in (1, 2, 3):
    _, _, tr = U.build(arrays, cam)
    tr.trace(d, flags=FLAGS, state=False)
    torch.cuda.synchronize()
    want = c_oracle.render(arrays, cam, d)
    got = tr.sample_buf.cpu().numpy().reshape(-1, 3)
    err = np.abs(got - want["sample"]).max(axis=1)
    solid = ~want["fragile"]
    bad = np.nonzero((err > 1e-4) & solid)[0]
    print(f"depth {d}: {bad.size} bad pixels of {err.size}; worst {err[solid].max():.3e}")
    if bad.size:
        i, j = bad // H, bad % H
        print("  i range", i.min(), i.max(), "j range", j.min(), j.max(), "tiles", sorted(set(zip((i // 16).tolist(), (j // 16).tolist())))[:20])
        for b in bad[:5]:
            print("  px", b // H, b % H, "got", got[b], "want", want["sample"][b], "hits", want["hits"][b])
    # the listed path
    _, _, tl = U.build(arrays, cam)
    for _ in range(d):
        tl.sample(d)
    torch.cuda.synchronize()
    gl = tl.sample_buf.cpu().numpy().reshape(-1, 3)
    el = np.abs(gl - want["sample"]).max(axis=1)
    print(f"  listed sample() x {d}: worst {el[solid].max():.3e}, bad {int(((el > 1e-4) & solid).sum())}")
