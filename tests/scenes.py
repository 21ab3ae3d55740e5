"""Deterministic test scenes shared by the golden generator and the tests (inputs only)."""

from __future__ import annotations

import math

import numpy as np

from rtgs import synthetic


def quat_axis_angle(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    axis = axis / np.linalg.norm(axis)
    s = math.sin(angle / 2)
    return tuple(float(np.float32(v)) for v in (axis[0] * s, axis[1] * s, axis[2] * s, math.cos(angle / 2)))


def f32_tuple(values):
    """Camera parameters cross the C ABI as fp32: keep test inputs exactly representable."""
    return tuple(float(np.float32(v)) for v in values)


def config_case(name):
    cfg = synthetic.CONFIGS[name]
    return synthetic.make_config_scene(cfg), synthetic.bench_camera(cfg.width, cfg.height), cfg.depth


def oblique_case():
    """Rotated camera, fx != fy, odd image size not a multiple of 32 or 8, N not a multiple of 64."""
    arrays = synthetic.make_scene(1500, 1.4, 2024)
    cam = dict(position=f32_tuple((0.7, -0.4, 2.6)), rotation=quat_axis_angle((0.3, 1.0, 0.2), 0.35),
               width=150, height=101, fx=120.0, fy=135.0)
    return arrays, cam, 16


def inside_case():
    """Camera INSIDE the cloud: Gaussians behind the camera, containing it, and crossing the
    camera plane (the general-conic path of k_project)."""
    arrays = synthetic.make_scene(800, 1.0, 77)
    arrays["scale"] = (arrays["scale"] * np.float32(2.5)).astype(np.float32)
    cam = dict(position=f32_tuple((0.1, 0.05, 0.2)), rotation=quat_axis_angle((1.0, 0.2, -0.4), 0.8),
               width=96, height=80, fx=70.0, fy=70.0)
    return arrays, cam, 16


def sh_case():
    """Degree-3 SH colour (the c5 shading path) at a size the oracle finishes quickly."""
    arrays = synthetic.make_scene(3000, 16 / 9, 1005, sh=True)
    cam = synthetic.bench_camera(192, 108)
    cam["rotation"] = quat_axis_angle((0.0, 1.0, 0.0), 0.15)
    cam["position"] = f32_tuple((0.5, 0.1, 2.9))
    return arrays, cam, 16


def shallow_case():
    """depth < 16 and a dense scene: the depth cut is exercised on nearly every pixel."""
    arrays = synthetic.make_scene(4000, 1.0, 31)
    arrays["scale"] = (arrays["scale"] * np.float32(1.6)).astype(np.float32)
    return arrays, synthetic.bench_camera(96, 96), 5


CASES = {"oblique": oblique_case, "inside": inside_case, "sh": sh_case, "shallow": shallow_case}


def crop_pixels(width, height, cw=64, ch=36, seed=7, extra=1024):
    """A contiguous cw x ch crop around the image centre plus `extra` seeded random pixels."""
    i0, j0 = width // 2 - cw // 2, height // 2 - ch // 2
    i, j = np.meshgrid(np.arange(i0, i0 + cw), np.arange(j0, j0 + ch), indexing="ij")
    crop = np.stack([i.ravel(), j.ravel()], axis=-1)
    rng = np.random.default_rng(seed)
    rnd = np.stack([rng.integers(0, width, extra), rng.integers(0, height, extra)], axis=-1)
    return np.concatenate([crop, rnd]).astype(np.int32)


def all_pixels(width, height):
    i, j = np.meshgrid(np.arange(width), np.arange(height), indexing="ij")
    return np.stack([i.ravel(), j.ravel()], axis=-1).astype(np.int32)


def fuzz_case(seed):
    """One seeded random (scene, camera, depth, shard count, mode) case of tools/fuzz_gpu.py: image
    sizes 1..300 x 1..200, 1..9000 Gaussians (incl. slab-boundary counts), scale multipliers
    0.3..12 with optional 3..20x needles, cameras from inside the cloud to 8 units away, any
    orientation, fx != fy, depth 1..16.  mode 0 = the reference's one-peel-per-call pattern."""
    rng = np.random.default_rng(seed)
    w, h = int(rng.integers(1, 300)), int(rng.integers(1, 200))
    n = int(rng.choice([1, 2, 63, 64, 65, 300, 2000, 9000]))
    sh = bool(rng.random() < 0.25)
    arrays = synthetic.make_scene(n, max(w / h, 0.2), seed, sh=sh)
    arrays["scale"] = (arrays["scale"] * np.float32(rng.choice([0.3, 1.0, 3.0, 12.0]))).astype(np.float32)
    if rng.random() < 0.3:
        arrays["scale"][:, int(rng.integers(0, 3))] *= np.float32(rng.uniform(3, 20))
    rot = quat_axis_angle(rng.normal(size=3), rng.uniform(0, 3.1))
    pos = f32_tuple(rng.normal(size=3) * rng.choice([0.2, 1.0, 3.0, 8.0]))
    f = float(np.float32(rng.uniform(0.3, 3.0) * max(h, 8)))
    cam = dict(position=pos, rotation=rot, width=w, height=h, fx=f, fy=float(np.float32(f * rng.uniform(0.7, 1.4))))
    depth = int(rng.integers(1, 17))
    world = int(rng.choice([1, 1, 2, 5, 8]))
    mode = int(rng.integers(0, 4))
    # (round 5, drawn last so that every earlier field of a seed is what it was) NON-UNIT quaternions, as the reference
    # takes them: Gaussian.init does not normalise (gaussian.py:59-84) and rot_vec3 of a non-unit camera quaternion
    # stretches the ray directions by |q|^2 (camera.py:52) -- the class of input behind round 5's horizon-bound bug
    if rng.random() < 0.3:
        cam["rotation"] = f32_tuple(np.asarray(cam["rotation"]) * rng.uniform(0.75, 1.35))
    if rng.random() < 0.3:
        arrays["rotation"] = (arrays["rotation"] * rng.uniform(0.7, 1.4, (arrays["rotation"].shape[0], 1))).astype(np.float32)
    return arrays, cam, depth, world, mode


def big_fuzz_case(seed):
    """tools/fuzz_big.py generator (round 5): what fuzz_case never reached -- scenes large enough for the per-frame
    slab BINS (>= 513 slabs = 32 832 Gaussians; k_bin, the per-tile lists, their overflow fall-backs) and for the
    two-level scan (> 4096 slabs), under random cameras from inside the cloud to 8 units away, any orientation, image
    sizes 1..260 x 1..160 (tile grids that are not whole supertiles), scale multipliers 0.3..3, needles, SH, non-unit
    quaternions, sharded traces.  Returns (arrays, cam, depth, world, pixels): the oracle is evaluated on `pixels`;
    world == 0 asks for the reference's one-peel-per-sample() pattern instead of a sharded trace."""
    rng = np.random.default_rng(seed)
    w, h = int(rng.integers(1, 261)), int(rng.integers(1, 161))
    n = int(rng.choice([32_832, 33_000, 50_000, 66_000, 140_000, 270_000]))
    sh = bool(rng.random() < 0.25)
    arrays = synthetic.make_scene(n, max(w / h, 0.2), seed, sh=sh)
    arrays["scale"] = (arrays["scale"] * np.float32(rng.choice([0.3, 1.0, 3.0]))).astype(np.float32)
    if rng.random() < 0.2:
        arrays["scale"][:, int(rng.integers(0, 3))] *= np.float32(rng.uniform(3, 10))
    rot = quat_axis_angle(rng.normal(size=3), rng.uniform(0, 3.1))
    pos = f32_tuple(rng.normal(size=3) * rng.choice([0.2, 1.0, 3.0, 8.0]))
    f = float(np.float32(rng.uniform(0.3, 3.0) * max(h, 8)))
    cam = dict(position=pos, rotation=rot, width=w, height=h, fx=f, fy=float(np.float32(f * rng.uniform(0.7, 1.4))))
    depth = int(rng.integers(1, 17))
    world = int(rng.choice([1, 1, 2, 5, 8]))
    if rng.random() < 0.2:
        cam["rotation"] = f32_tuple(np.asarray(cam["rotation"]) * rng.uniform(0.75, 1.35))
    if rng.random() < 0.2:
        arrays["rotation"] = (arrays["rotation"] * rng.uniform(0.7, 1.4, (n, 1))).astype(np.float32)
    k = min(w * h, 900)
    flat = rng.choice(w * h, k, replace=False)
    pixels = np.stack([flat // h, flat % h], axis=-1).astype(np.int32)
    # (drawn last) one case in four in the reference's call pattern: `depth` sample() calls of one peel each -- the
    # listing instantiation of k_trace and k_peel instead of the compositing one
    if rng.random() < 0.25:
        world = 0
    return arrays, cam, depth, world, pixels


def needle_case(seed):
    """tools/fuzz_needles.py generator: capture-like Gaussians, log-uniform scales, anisotropy up
    to ~300:1 (sub-pixel-wide needles), random camera.  Returns (arrays, cam)."""
    rng = np.random.default_rng(seed)
    w, h = int(rng.integers(64, 200)), int(rng.integers(48, 160))
    n = int(rng.choice([500, 3000, 12000]))
    arrays = synthetic.make_scene(n, w / h, seed)
    lo, hi = np.log(1e-3), np.log(1e-1)
    base = np.exp(rng.uniform(lo, hi, (n, 1)))
    aniso = np.exp(rng.uniform(0, np.log(rng.choice([10.0, 100.0, 300.0])), (n, 3)))
    arrays["scale"] = (base * aniso / aniso.max(axis=1, keepdims=True) * rng.choice([1.0, 3.0])).astype(np.float32)
    cam = synthetic.bench_camera(w, h)
    cam["position"] = f32_tuple(rng.normal(size=3) * 0.3 + np.array([0, 0, rng.choice([1.5, 3.0, 6.0])]))
    cam["rotation"] = quat_axis_angle(rng.normal(size=3), rng.uniform(0, 0.4))
    cam["fx"] = cam["fy"] = float(np.float32(h * rng.uniform(0.8, 2.5)))
    return arrays, cam
