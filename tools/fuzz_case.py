import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "rt-gaussian-splat-renderer_amd"), ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import rtgs, scenes
from rtgs import synthetic
from oracle import c_oracle, rtgs_ref as ref

def make(seed):
    arrays, cam, depth, _, _ = scenes.fuzz_case(seed)
    return arrays, cam, depth

seed = int(sys.argv[1])
arrays, cam, depth = make(seed)
w, h = cam["width"], cam["height"]
print("seed", seed, "n", arrays["position"].shape[0], "wxh", w, h, "depth", depth, "cam", cam["position"], "scale max", arrays["scale"].max(), "min", arrays["scale"].min())
scene = rtgs.Scene.from_arrays(**arrays)
camera = rtgs.Camera(cam["position"], cam["rotation"], (w, h), (cam["fx"], cam["fy"]))
tr = rtgs.RayTracer((w, h), scene, camera)
tr.trace(depth); torch.cuda.synchronize()
got = tr.sample_buf.cpu().numpy().reshape(-1, 3)
want = c_oracle.render(arrays, cam, depth)
err = np.abs(got - want["sample"]).max(axis=1); solid = ~want["fragile"]
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-4
bad = np.nonzero((err > thr) & solid)[0]
print("bad robust pixels", bad.size, "of", solid.sum())
osc = ref.OracleScene(**arrays)
for p in bad[np.argsort(-err[bad])][:4]:
    pix = np.array([[p // h, p % h]])
    r = ref.render_topk(osc, cam["position"], cam["rotation"], w, h, cam["fx"], cam["fy"], depth=depth, pixels=pix)
    t = r["t1"][0]; t = t[np.isfinite(t)]
    print(" pixel", pix[0], "err %.3e" % err[p], "hits", r["hits"][0], "margin_q %.3e margin_t %.3e" % (r["margin_q"][0], r["margin_t"][0]))
    print("   t1:", np.round(t, 6).tolist())
    print("   gaps:", ["%.1e" % g for g in np.diff(t)])
    ids = r["idx"][0]; ids = ids[ids >= 0]
    sub = {k: (None if v is None else np.ascontiguousarray(v[ids])) for k, v in arrays.items()}
    sc2 = rtgs.Scene.from_arrays(**sub); tr2 = rtgs.RayTracer((w, h), sc2, camera); tr2.trace(depth); torch.cuda.synchronize()
    gp = tr2.sample_buf[pix[0, 0], pix[0, 1]].cpu().numpy()
    print("   GPU on oracle's set:", gp, "want", want["sample"][p], "full-scene GPU", got[p])
