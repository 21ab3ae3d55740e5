"""Dump GPU and C-oracle images of one tools/fuzz_needles.py case to gpurun_out/needle_<seed>.npz."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "rt-gaussian-splat-renderer_amd"), ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import rtgs, scenes
from rtgs import synthetic
from oracle import c_oracle


needle_case = scenes.needle_case


if __name__ == "__main__":
    seed = int(sys.argv[1])
    arrays, cam = needle_case(seed)
    w, h = cam["width"], cam["height"]
    scene = rtgs.Scene.from_arrays(**arrays)
    camera = rtgs.Camera(cam["position"], cam["rotation"], (w, h), (cam["fx"], cam["fy"]))
    tr = rtgs.RayTracer((w, h), scene, camera); tr.trace(16); torch.cuda.synchronize()
    got = tr.sample_buf.cpu().numpy().reshape(-1, 3)
    want = c_oracle.render(arrays, cam, 16)
    err = np.abs(got - want["sample"]).max(axis=1); solid = ~want["fragile"]
    k = int(np.argmax(np.where(solid, err, 0)))
    print(f"seed {seed}: {w}x{h}, worst robust pixel {k} = ({k // h}, {k % h}) err {err[k]:.3e}; got {got[k]} want {want['sample'][k]}")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.savez(os.path.join(ROOT, "gpurun_out", f"needle_{seed}.npz"), got=got, want=want["sample"], fragile=want["fragile"],
             start=tr.camera.cam_ray_field.start.cpu().numpy().reshape(-1), att=tr.attenuation_buf.cpu().numpy().reshape(-1))
