"""Seeded synthetic random-Gaussian scenes and the benchmark camera.

Host-side NumPy only (scene construction, like the reference's loader scene.py:95-114 is host
NumPy code).  The five BASELINE.json configurations are named c1..c5.
"""

from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np


@dataclass(frozen=True)
class Config:
    name: str
    n: int
    width: int
    height: int
    seed: int
    sh: bool = False
    depth: int = 16

    @property
    def aspect(self) -> float:
        return self.width / self.height

    @property
    def rays(self) -> int:
        return self.width * self.height


CONFIGS = {
    "c1": Config("c1", 256, 128, 128, 1001),
    "c2": Config("c2", 10_000, 800, 600, 1002),
    "c3": Config("c3", 100_000, 1920, 1080, 1003),
    "c4": Config("c4", 100_000, 1920, 1080, 1003),       # c3 sharded over 8 GPUs
    "c5": Config("c5", 1_000_000, 3840, 2160, 1005, sh=True),
    # not a BASELINE config: c3's size with capture-like statistics, see make_capture_scene
    "cap": Config("cap", 100_000, 1920, 1080, 1013),
    # not a BASELINE config: c3 with degree-3 SH (the kernel a PLY capture with f_rest takes)
    "c3sh": Config("c3sh", 100_000, 1920, 1080, 1003, sh=True),
}


def bytes_per_ray(n: int, sh: bool = False, depth: int = 16) -> int:
    """ALGORITHMIC bytes per ray fixed by SURVEY.md 8(d): every 64-pixel wave tile streams
    every Gaussian's 36-B geometry, shading data for `depth` hits, one framebuffer write."""
    return 36 * n // 64 + depth * 16 + (depth * 180 if sh else 0) + 12


def make_scene(n: int, aspect: float, seed: int, sh: bool = False) -> dict:
    """Random Gaussians in the box [-aspect, aspect] x [-1, 1] x [-1, 1].

    Only rng.random() is used, in this fixed order: position (n,3), quaternion uniforms (n,3),
    scale exponents (n,3), color (n,3), opacity (n), [sh (n,15,3)].  Everything is cast to fp32.
    """
    rng = np.random.default_rng(seed)
    pos = rng.random((n, 3)) * 2.0 - 1.0
    pos[:, 0] *= aspect
    u = rng.random((n, 3))
    # Shoemake: uniform unit quaternion from three uniforms, stored scalar-last (x, y, z, w)
    a, b = np.sqrt(1.0 - u[:, 0]), np.sqrt(u[:, 0])
    t1, t2 = 2.0 * math.pi * u[:, 1], 2.0 * math.pi * u[:, 2]
    rot = np.stack([a * np.sin(t1), a * np.cos(t1), b * np.sin(t2), b * np.cos(t2)], axis=-1)
    s0 = math.sqrt(8.727 * aspect / n)
    scale = s0 * np.exp2(rng.random((n, 3)) * 2.0 - 1.0)
    color = rng.random((n, 3))
    opacity = 0.1 + 0.8 * rng.random(n)
    out = dict(position=pos, rotation=rot, scale=scale, color=color, opacity=opacity, sh=None)
    if sh:
        out["sh"] = rng.random((n, 15, 3)) * 0.1 - 0.05
    return {k: (None if v is None else np.ascontiguousarray(v, dtype=np.float32))
            for k, v in out.items()}


def make_capture_scene(n: int, aspect: float, seed: int, sigma: float = 1.2, background: float = 0.01) -> dict:
    """A scene with the size statistics of a captured 3DGS model (the reference's use case is
    truck.ply, ~150 k Gaussians: docs/source/get-started.md:64-74) instead of c3's 4:1 spread:
    per-Gaussian size log-normal (sigma in ln s), per-axis anisotropy on top, and `background`
    (1 %) of the Gaussians large (6 x s0), faint and in the back third of the box.  The median
    size is chosen so that the log-normal part covers the screen as c3 does (E[s^2] = s0^2).
    Only rng.random() is used (Box-Muller for the normals)."""
    rng = np.random.default_rng(seed)
    base = make_scene(n, aspect, seed)
    u = rng.random((n, 2))
    z = np.sqrt(-2.0 * np.log(1.0 - u[:, 0])) * np.cos(2.0 * math.pi * u[:, 1])     # N(0, 1)
    s0 = math.sqrt(8.727 * aspect / n)
    size = s0 * math.exp(-sigma * sigma) * np.exp(sigma * z)                          # median s0 e^{-sigma^2}
    scale = size[:, None] * np.exp2(rng.random((n, 3)) * 2.0 - 1.0)
    pos = base["position"].astype(np.float64)
    opacity = base["opacity"].astype(np.float64)
    nb = int(round(n * background))
    if nb:
        pick = rng.random(n).argsort()[:nb]
        scale[pick] = 6.0 * s0 * np.exp2(rng.random((nb, 3)) - 0.5)
        pos[pick, 2] = -1.0 + (2.0 / 3.0) * rng.random(nb)
        opacity[pick] = 0.05 + 0.25 * rng.random(nb)
    out = dict(base, position=pos, scale=scale, opacity=opacity)
    return {k: (None if v is None else np.ascontiguousarray(v, dtype=np.float32)) for k, v in out.items()}


def make_config_scene(cfg: Config | str) -> dict:
    if isinstance(cfg, str):
        cfg = CONFIGS[cfg]
    if cfg.name == "cap":
        return make_capture_scene(cfg.n, cfg.aspect, cfg.seed)
    return make_scene(cfg.n, cfg.aspect, cfg.seed, cfg.sh)


def bench_camera(width: int, height: int) -> dict:
    """Camera of SURVEY.md 8(d): at (0,0,3) looking down -z, +y up, focal = H pixels (vertical
    fov 53.13 deg), so the box's front face z = +1 exactly fills the frame."""
    return dict(position=(0.0, 0.0, 3.0), rotation=(0.0, 0.0, 0.0, 1.0), width=width,
                height=height, fx=float(height), fy=float(height))
