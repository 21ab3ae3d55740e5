"""CPU oracle: a NumPy restatement of the reference ray-traced Gaussian-splat renderer.

THIS FILE IS TEST INFRASTRUCTURE.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product (``rtgs`` package + HIP
library) never does.

It restates, function by function, the algorithm of fangjunzhou/rt-gaussian-splat-renderer
(reference @ 2025-06-29); every function cites the reference ``file:line`` it follows.  The
reference runs these as Taichi ``@ti.func``/``@ti.kernel`` code in fp32; here every function
takes a ``dtype`` (``np.float32`` = the reference's arithmetic type, ``np.float64`` = the
golden used for pixel parity).

PINNING STATUS
  * pinned by the reference's own tests (tests/test_oracle_kat.py): quaternion ``mul`` /
    ``conj`` / axis-angle / ``rot_vec3`` / ``as_rotation_mat3|4`` (tests/test_quaternion.py),
    ``Ray.get`` and the Ray / Gaussian defaults (tests/test_ray.py, tests/test_gaussian.py),
    and the ``tests/data/test.ply`` fixture decode.
  * PARITY UNPINNED for ``Gaussian.hit / eval / eval_sh``, ``Camera.generate_ray*``,
    ``Scene.hit`` and ``RayTracer.sample*``: the reference has no test that pins a value for
    them (its one ``Gaussian.hit`` expectation contradicts its own arithmetic),
    and the reference itself needs taichi==1.7.3, which is absent from this image, so it
    cannot be executed to generate vectors.  Those functions are pinned only by the analytic
    known-answer cases in tests/test_oracle_analytic.py, derived by hand from the cited lines.
"""

from __future__ import annotations

import math

import numpy as np

# gaussian.py:13 -- squared Mahalanobis radius of the hit ellipsoid.
BOUNDING_THRESHOLD = 3

# gaussian.py:17-23 -- spherical-harmonics constants.
C_0 = math.sqrt(3 / math.pi)
C_1 = math.sqrt(15 / math.pi)
C_2 = math.sqrt(5 / math.pi)
C_3 = math.sqrt(35 / (2 * math.pi))
C_4 = math.sqrt(105 / math.pi)
C_5 = math.sqrt(21 / (2 * math.pi))
C_6 = math.sqrt(7 / math.pi)

# ray_tracer.py:100
EPS = 1e-8


def _a(x, dtype):
    return np.asarray(x, dtype=dtype)


# --------------------------------------------------------------------------------------
# utils/quaternion.py -- scalar-last (x, y, z, w) quaternions
# --------------------------------------------------------------------------------------

def quat_mul(p, q, dtype=np.float64):
    """utils/quaternion.py:8-23 -- Hamilton product pq, scalar last."""
    p = _a(p, dtype)
    q = _a(q, dtype)
    pv, pw = p[..., :3], p[..., 3:4]
    qv, qw = q[..., :3], q[..., 3:4]
    w = pw * qw - np.sum(pv * qv, axis=-1, keepdims=True)
    v = pw * qv + qw * pv + np.cross(pv, qv)
    return np.concatenate([v, w], axis=-1).astype(dtype)


def quat_conj(q, dtype=np.float64):
    """utils/quaternion.py:26-35."""
    q = _a(q, dtype)
    return np.concatenate([-q[..., :3], q[..., 3:4]], axis=-1)


def quat_inv(q, dtype=np.float64):
    """utils/quaternion.py:38-47 -- literal: conj(q) / |q| (only an inverse for unit q)."""
    q = _a(q, dtype)
    return quat_conj(q, dtype) / np.sqrt(np.sum(q * q, axis=-1, keepdims=True))


def quat_from_axis_angle(v, dtype=np.float64):
    """utils/quaternion.py:50-65."""
    v = _a(v, dtype)
    theta = np.sqrt(np.sum(v * v, axis=-1, keepdims=True))
    safe = np.where(theta > 0, theta, 1)
    axis = np.where(theta > 0, v / safe * np.sin(theta / 2), v)
    w = np.cos(theta / 2)
    return np.concatenate([axis, w], axis=-1).astype(dtype)


def quat_as_axis_angle(q, dtype=np.float64):
    """utils/quaternion.py:68-81."""
    q = _a(q, dtype)
    theta = np.arccos(q[..., 3:4]) * 2
    norm = np.sqrt(np.sum(q[..., :3] ** 2, axis=-1, keepdims=True))
    safe = np.where(norm > 0, norm, 1)
    return np.where(norm > 0, q[..., :3] / safe * theta, 0).astype(dtype)


def quat_rot_vec3(q, v, dtype=np.float64):
    """utils/quaternion.py:84-96 -- (q (v,0) q*).xyz."""
    q = _a(q, dtype)
    v = _a(v, dtype)
    qv = np.concatenate([v, np.zeros(v.shape[:-1] + (1,), dtype=dtype)], axis=-1)
    return quat_mul(q, quat_mul(qv, quat_conj(q, dtype), dtype), dtype)[..., :3]


def quat_as_rotation_mat3(q, dtype=np.float64):
    """utils/quaternion.py:99-121 -- column k is q e_k q*."""
    q = _a(q, dtype)
    cols = []
    for k in range(3):
        e = np.zeros(q.shape[:-1] + (3,), dtype=dtype)
        e[..., k] = 1
        cols.append(quat_rot_vec3(q, e, dtype))
    return np.stack(cols, axis=-1)  # [..., row, col]


def quat_as_rotation_mat4(q, dtype=np.float64):
    """utils/quaternion.py:124-147 -- mat3 embedded in a 4x4 identity."""
    m3 = quat_as_rotation_mat3(q, dtype)
    m = np.zeros(m3.shape[:-2] + (4, 4), dtype=dtype)
    m[..., :3, :3] = m3
    m[..., 3, 3] = 1
    return m


# --------------------------------------------------------------------------------------
# ray.py
# --------------------------------------------------------------------------------------

RAY_DEFAULT = dict(origin=(0.0, 0.0, 0.0), direction=(0.0, 1.0, 0.0), start=0.0, end=math.inf)
"""ray.py:20-41 / 55-68 -- Ray.init / new_ray defaults."""

GAUSSIAN_DEFAULT = dict(position=(0.0, 0.0, 0.0), rotation=(0.0, 0.0, 0.0, 1.0),
                        scale=(1.0, 1.0, 1.0), color=(1.0, 0.0, 1.0), opacity=1.0)
"""gaussian.py:59-84 / 233-247 -- Gaussian.init / new_gaussian defaults."""


def ray_get(origin, direction, t, dtype=np.float64):
    """ray.py:43-52 -- o + t d."""
    return _a(origin, dtype) + _a(t, dtype)[..., None] * _a(direction, dtype)


# --------------------------------------------------------------------------------------
# camera.py
# --------------------------------------------------------------------------------------

def all_pixels(width, height):
    """(W*H, 2) int array of (i, j), i major -- the (W, H) field order of camera.py:29."""
    i, j = np.meshgrid(np.arange(width), np.arange(height), indexing="ij")
    return np.stack([i.ravel(), j.ravel()], axis=-1)


def generate_rays(position, rotation, width, height, fx, fy, pixels=None, dtype=np.float64):
    """camera.py:31-71 -- pinhole rays.  Returns (origin (3,), dirs (n, 3)).

    uv = ((i+.5)/W, (j+.5)/H); pxy = (sensor*uv - .5*sensor)/focal with sensor = buf_size;
    d_cam = normalize(pxy.x, pxy.y, -1); d = rot_vec3(q, d_cam).
    """
    if pixels is None:
        pixels = all_pixels(width, height)
    pixels = np.asarray(pixels)
    one = dtype(1)
    u = (pixels[:, 0].astype(dtype) + dtype(0.5)) / dtype(width)
    v = (pixels[:, 1].astype(dtype) + dtype(0.5)) / dtype(height)
    px = (dtype(width) * u - dtype(0.5) * dtype(width)) / dtype(fx)
    py = (dtype(height) * v - dtype(0.5) * dtype(height)) / dtype(fy)
    d_cam = np.stack([px, py, -np.ones_like(px)], axis=-1)
    d_cam = d_cam / np.sqrt(np.sum(d_cam * d_cam, axis=-1, keepdims=True))
    d = quat_rot_vec3(_a(rotation, dtype)[None, :], d_cam, dtype)
    return _a(position, dtype), d.astype(dtype) * one


# --------------------------------------------------------------------------------------
# gaussian.py
# --------------------------------------------------------------------------------------

def gaussian_cov(rotation, scale, dtype=np.float64):
    """gaussian.py:86-102 -- R S S^T R^T."""
    r = quat_as_rotation_mat3(rotation, dtype)
    s = _a(scale, dtype)
    smat = np.zeros(s.shape[:-1] + (3, 3), dtype=dtype)
    for k in range(3):
        smat[..., k, k] = s[..., k]
    rs = r @ smat
    return (rs @ np.swapaxes(smat, -1, -2)) @ np.swapaxes(r, -1, -2)


def gaussian_bounding_box(position, rotation, scale, threshold=3.0, dtype=np.float64):
    """gaussian.py:104-138 -- the AABB of the six points position +- R (S (threshold e_k)), k = x, y, z: the end
    points of the ellipsoid's axes at `threshold` (BOUNDING_THRESHOLD = 3, gaussian.py:13) scale units, NOT the
    ellipsoid's tight box (the reference's own TODO, :112).  Returns (p_min, p_max)."""
    r = quat_as_rotation_mat3(rotation, dtype)
    s = _a(scale, dtype)
    p = _a(position, dtype)
    pts = []
    for k in range(3):
        e = np.zeros(s.shape[:-1] + (3,), dtype=dtype)
        e[..., k] = threshold
        arm = (r @ (s * e)[..., None])[..., 0]
        pts += [p + arm, p - arm]
    pts = np.stack(pts, axis=0)
    return pts.min(axis=0), pts.max(axis=0)


def mat3_inverse(m, dtype=np.float64):
    """ti.math.inverse on a 3x3 (gaussian.py:196, 215): adjugate / determinant, in `dtype`."""
    m = _a(m, dtype)
    a, b, c = m[..., 0, 0], m[..., 0, 1], m[..., 0, 2]
    d, e, f = m[..., 1, 0], m[..., 1, 1], m[..., 1, 2]
    g, h, i = m[..., 2, 0], m[..., 2, 1], m[..., 2, 2]
    co00 = e * i - f * h
    co01 = -(d * i - f * g)
    co02 = d * h - e * g
    det = a * co00 + b * co01 + c * co02
    inv = np.empty_like(m)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv[..., 0, 0] = co00 / det
        inv[..., 1, 0] = co01 / det
        inv[..., 2, 0] = co02 / det
        inv[..., 0, 1] = -(b * i - c * h) / det
        inv[..., 1, 1] = (a * i - c * g) / det
        inv[..., 2, 1] = -(a * h - b * g) / det
        inv[..., 0, 2] = (b * f - c * e) / det
        inv[..., 1, 2] = -(a * f - c * d) / det
        inv[..., 2, 2] = (a * e - b * d) / det
    return inv


def gaussian_cov_inv(rotation, scale, dtype=np.float64):
    """inverse(cov()) as used by gaussian.py:196 and :215."""
    return mat3_inverse(gaussian_cov(rotation, scale, dtype), dtype)


def _hit_from_abc(A, B, C, dtype):
    """gaussian.py:219-230 -- roots of A t^2 + B t + C."""
    delta = B * B - dtype(4) * A * C
    inf = dtype(np.inf)
    with np.errstate(divide="ignore", invalid="ignore"):
        sq = np.sqrt(np.where(delta > 0, delta, 0))
        t1 = np.where(delta > 0, (-B - sq) / (2 * A), np.where(delta == 0, -B / (2 * A), inf))
        t2 = np.where(delta > 0, (-B + sq) / (2 * A), inf)
    return t1.astype(dtype), t2.astype(dtype), delta


def gaussian_hit(origin, direction, position, cov_inv, dtype=np.float64):
    """gaussian.py:203-230, elementwise over broadcastable ray / Gaussian batches.

    origin, direction: (..., 3); position: (..., 3); cov_inv: (..., 3, 3).
    Returns (t1, t2), both +inf on a miss.
    """
    o = _a(origin, dtype)
    d = _a(direction, dtype)
    mu = _a(position, dtype)
    M = _a(cov_inv, dtype)
    m = o - mu
    Md = np.einsum("...ij,...j->...i", M, d)
    Mm = np.einsum("...ij,...j->...i", M, m)
    A = np.sum(d * Md, axis=-1)
    B = 2 * np.sum(d * Mm, axis=-1)
    C = np.sum(m * Mm, axis=-1) - dtype(BOUNDING_THRESHOLD)
    t1, t2, _ = _hit_from_abc(A, B, C, dtype)
    return t1, t2


def sh_basis(direction, dtype=np.float64):
    """gaussian.py:149-163 -- the 15 degree-1..3 basis values, INCLUDING the y_33 quirk
    (5 z^2 - 3 z, not 5 z^3 - 3 z).  direction must already be normalised.  (..., 15)."""
    d = _a(direction, dtype)
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    f = dtype
    ys = [
        f(0.5 * C_0) * y,
        f(0.5 * C_0) * z,
        f(0.5 * C_0) * x,
        f(0.5 * C_1) * x * y,
        f(0.5 * C_1) * y * z,
        f(0.25 * C_2) * (3 * z ** 2 - 1),
        f(0.5 * C_1) * x * z,
        f(0.25 * C_1) * (x ** 2 - y ** 2),
        f(0.25 * C_3) * y * (3 * x ** 2 - y ** 2),
        f(0.5 * C_4) * x * y * z,
        f(0.25 * C_5) * y * (5 * z ** 2 - 1),
        f(0.25 * C_6) * (5 * z ** 2 - 3 * z),
        f(0.25 * C_5) * x * (5 * z ** 2 - 1),
        f(0.25 * C_4) * (x ** 2 - y ** 2) * z,
        f(0.25 * C_3) * x * (x ** 2 - 3 * y ** 2),
    ]
    return np.stack(ys, axis=-1).astype(dtype)


def gaussian_eval_sh(direction, sh, dtype=np.float64):
    """gaussian.py:140-181 -- sum_k Y_k(dir) * sh_k; sh: (..., 15, 3)."""
    return np.einsum("...k,...kc->...c", sh_basis(direction, dtype), _a(sh, dtype))


def gaussian_eval(pos, direction, position, cov_inv, color, opacity, sh=None, dtype=np.float64):
    """gaussian.py:183-201 -- returns (rgb (..., 3), alpha (...)).

    rho = exp(-(pos-mu)^T Sigma^-1 (pos-mu))  (no 1/2), alpha = opacity*rho,
    rgb = color + eval_sh(normalize(dir)); no clamp."""
    x = _a(pos, dtype) - _a(position, dtype)
    M = _a(cov_inv, dtype)
    with np.errstate(invalid="ignore", over="ignore"):
        q = np.sum(x * np.einsum("...ij,...j->...i", M, x), axis=-1)
        rho = np.exp(-q)
    alpha = _a(opacity, dtype) * rho
    rgb = _a(color, dtype) * np.ones(alpha.shape + (1,), dtype=dtype)
    if sh is not None:
        d = _a(direction, dtype)
        d = d / np.sqrt(np.sum(d * d, axis=-1, keepdims=True))
        rgb = rgb + gaussian_eval_sh(d, sh, dtype)
    return rgb.astype(dtype), alpha.astype(dtype)


# --------------------------------------------------------------------------------------
# scene.py -- closest-hit semantics (the BVH is only an accelerator of this query)
# --------------------------------------------------------------------------------------

class OracleScene:
    """Post-activation Gaussian arrays (what Gaussian.init takes, gaussian.py:59-84) plus the
    per-Gaussian inverse covariance, hoisted out of the per-ray loop (the reference recomputes
    it per ray per Gaussian, gaussian.py:196,215; same arithmetic, same result)."""

    def __init__(self, position, rotation, scale, color, opacity, sh=None, dtype=np.float64):
        self.dtype = dtype
        self.position = _a(position, dtype).reshape(-1, 3)
        self.rotation = _a(rotation, dtype).reshape(-1, 4)
        self.scale = _a(scale, dtype).reshape(-1, 3)
        self.color = _a(color, dtype).reshape(-1, 3)
        self.opacity = _a(opacity, dtype).reshape(-1)
        self.sh = None if sh is None else _a(sh, dtype).reshape(-1, 15, 3)
        self.n = self.position.shape[0]
        self.cov_inv = gaussian_cov_inv(self.rotation, self.scale, dtype)

    def astype(self, dtype):
        return OracleScene(self.position, self.rotation, self.scale, self.color, self.opacity,
                           self.sh, dtype)


def _pair_t(scene, origin, dirs, g0, g1):
    """(t1, t2, delta/(4A), -B/(2A)) of rays x Gaussians[g0:g1] (gaussian.py:203-230);
    origin may be (3,) or (n, 3)."""
    dt = scene.dtype
    M = scene.cov_inv[g0:g1]                       # (G,3,3)
    mu = scene.position[g0:g1]                     # (G,3)
    o = np.broadcast_to(origin, dirs.shape)
    if origin.ndim == 1:
        m = origin[None, :] - mu                   # (G,3)
        Mm = np.einsum("gij,gj->gi", M, m)
        Bh = dirs @ Mm.T                           # (R,G)  = d^T M m
        C = (np.sum(m * Mm, axis=-1) - dt(BOUNDING_THRESHOLD))[None, :]
    else:
        m = o[:, None, :] - mu[None, :, :]         # (R,G,3)
        Mm = np.einsum("gij,rgj->rgi", M, m)
        Bh = np.sum(dirs[:, None, :] * Mm, axis=-1)
        C = np.sum(m * Mm, axis=-1) - dt(BOUNDING_THRESHOLD)
    Md = np.einsum("gij,rj->rgi", M, dirs)
    A = np.sum(dirs[:, None, :] * Md, axis=-1)
    B = 2 * Bh
    t1, t2, delta = _hit_from_abc(A, B, C, dt)
    with np.errstate(divide="ignore", invalid="ignore"):
        w = delta / (4 * A)
        tc = -B / (2 * A)
    return t1, t2, w, tc


def closest_hit(scene, origin, dirs, start, end, chunk=None):
    """scene.py:406-450 (criteria :433-437): argmin t1 s.t. start < t1 < end, strict '<' so the
    lowest index wins exact ties here.  Returns (idx (-1 = miss), t1, t2)."""
    dt = scene.dtype
    origin = _a(origin, dt)
    dirs = _a(dirs, dt)
    n = dirs.shape[0]
    start = np.broadcast_to(_a(start, dt), (n,))
    end = np.broadcast_to(_a(end, dt), (n,))
    best_t = np.full(n, np.inf, dtype=dt)
    best_t2 = np.full(n, np.inf, dtype=dt)
    best_i = np.full(n, -1, dtype=np.int64)
    if chunk is None:
        chunk = max(1, min(scene.n, (1 << 21) // max(n, 1)))
    rows = np.arange(n)
    for g0 in range(0, scene.n, chunk):
        g1 = min(scene.n, g0 + chunk)
        t1, t2, _, _ = _pair_t(scene, origin, dirs, g0, g1)
        ok = (t1 < end[:, None]) & (t1 > start[:, None])
        t1m = np.where(ok, t1, np.inf)
        k = np.argmin(t1m, axis=1)
        tk = t1m[rows, k]
        better = tk < best_t
        best_t = np.where(better, tk, best_t)
        best_t2 = np.where(better, t2[rows, k], best_t2)
        best_i = np.where(better, k + g0, best_i)
    return best_i, best_t, best_t2


# --------------------------------------------------------------------------------------
# ray_tracer.py -- depth peeling + front-to-back compositing
# --------------------------------------------------------------------------------------

def _shade(scene, origin, dirs, idx, t1, t2):
    """ray_tracer.py:88-94 -- evaluate Gaussian idx at the chord midpoint."""
    dt = scene.dtype
    with np.errstate(invalid="ignore"):
        t_mid = (t1 + t2) / dt(2)
        pos = origin + t_mid[:, None] * dirs
    sh = None if scene.sh is None else scene.sh[idx]
    return gaussian_eval(pos, dirs, scene.position[idx], scene.cov_inv[idx], scene.color[idx],
                         scene.opacity[idx], sh, dt)


def render_peel(scene, cam_position, cam_rotation, width, height, fx, fy, depth=16,
                pixels=None, state=None, steps=None, absorb_eps=None, t_cutoff=0.0):
    """The reference's literal loop (ray_tracer.py:39-54, 79-104): `depth` closest-hit peels,
    each evaluating the hit Gaussian at its chord midpoint and advancing ray.start.

    Returns dict(sample (n,3), attenuation (n,), start (n,), hits (n,) int).  `state` (a
    previous return value) resumes a partially peeled sample; `steps` limits the number of
    peels performed by this call.

    absorb_eps: ray.start = t1 + 1e-8 (ray_tracer.py:100-102).  In fp32 the 1e-8 is absorbed
    for t1 >~ 0.17, so the next peel's strict `t1 > start` (scene.py:433) is what excludes the
    same Gaussian.  The fp64 golden keeps that *observable* behaviour by advancing start to
    t1 exactly (absorb_eps=True, default for float64); float32 applies the literal addition.

    t_cutoff > 0 (an EXTENSION of the build, the reference has none: ray_tracer.py:79-104): a ray
    whose attenuation has fallen below t_cutoff is terminated -- it composites nothing more and
    its start becomes +inf, exactly as after a miss.
    """
    dt = scene.dtype
    if absorb_eps is None:
        absorb_eps = dt == np.float64
    origin, dirs = generate_rays(cam_position, cam_rotation, width, height, fx, fy, pixels, dt)
    n = dirs.shape[0]
    if state is None:
        state = dict(sample=np.zeros((n, 3), dtype=dt), attenuation=np.ones(n, dtype=dt),
                     start=np.zeros(n, dtype=dt), hits=np.zeros(n, dtype=np.int64))
    sample = state["sample"].copy()
    att = state["attenuation"].copy()
    start = state["start"].copy()
    hits = state["hits"].copy()
    inf = dt(np.inf)
    for _ in range(depth if steps is None else steps):
        idx, t1, t2 = closest_hit(scene, origin, dirs, start, inf)
        hit = idx >= 0
        if t_cutoff > 0:
            hit &= ~(att < dt(t_cutoff))
        if hit.any():
            h = np.nonzero(hit)[0]
            rgb, alpha = _shade(scene, origin, dirs[h], idx[h], t1[h], t2[h])
            sample[h] += (att[h] * alpha)[:, None] * rgb
            att[h] *= dt(1) - alpha
            start[h] = t1[h] if absorb_eps else (t1[h] + dt(EPS)).astype(dt)
            hits[h] += 1
        start[~hit] = inf
    return dict(sample=sample, attenuation=att, start=start, hits=hits)


def sample_step_rays(scene, rays, sample, attenuation, absorb_eps=None):
    """ray_tracer.py:79-104 on an ARBITRARY ray field, literally: rays (n, 8) = origin, direction,
    start, end per ray; returns the updated (rays, sample, attenuation) copies.  One peel."""
    dt = scene.dtype
    if absorb_eps is None:
        absorb_eps = dt == np.float64
    rays = _a(rays, dt).reshape(-1, 8).copy()
    sample = _a(sample, dt).reshape(-1, 3).copy()
    att = _a(attenuation, dt).reshape(-1).copy()
    origin, dirs = rays[:, 0:3], rays[:, 3:6]
    idx, t1, t2 = closest_hit(scene, origin, dirs, rays[:, 6], rays[:, 7])
    hit = idx >= 0
    if hit.any():
        h = np.nonzero(hit)[0]
        rgb, alpha = _shade(scene, origin[h], dirs[h], idx[h], t1[h], t2[h])
        sample[h] += (att[h] * alpha)[:, None] * rgb
        att[h] *= dt(1) - alpha
        rays[h, 6] = t1[h] if absorb_eps else (t1[h] + dt(EPS)).astype(dt)
    rays[~hit, 6] = dt(np.inf)
    return rays, sample, att


def render_topk(scene, cam_position, cam_rotation, width, height, fx, fy, depth=16,
                pixels=None, chunk=None, eps_q=1e-5, eps_t=1e-6, t_cutoff=0.0, eps_c=1e-6):
    """SURVEY.md Appendix A.9 -- the one-pass form of the same image: the <= depth Gaussians
    with the smallest entry distance t1 > 0, sorted ascending, composited front to back, with
    alpha = opacity * exp(delta/(4A) - 3) (== the chord-midpoint evaluation of gaussian.py:197).

    Also reports, per pixel, how close the result is to a discontinuity of the image function
    (a grazing hit, or two entry distances about to swap / cross the depth cut):
      margin_q = min |delta/(4A)| over Gaussians in front of the camera,
      margin_t = min gap between consecutive sorted t1 among the first depth+1 hits, relative
                 to max(1, t1) (an fp32 t1 carries a few ulp of error, which grows with t1).
    `fragile` = margin_q < eps_q or margin_t < eps_t: an fp32 renderer may legitimately take
    the other branch there.

    t_cutoff > 0 (extension, see render_peel): a ray stops compositing once its attenuation is
    below t_cutoff.  margin_c = the closest any attenuation on the way came to t_cutoff, relative;
    below eps_c the pixel is fragile too (an fp32 attenuation may fall on the other side).
    """
    dt = scene.dtype
    origin, dirs = generate_rays(cam_position, cam_rotation, width, height, fx, fy, pixels, dt)
    n = dirs.shape[0]
    K = depth + 1
    keep_t = np.full((n, K), np.inf, dtype=dt)
    keep_w = np.zeros((n, K), dtype=dt)
    keep_i = np.full((n, K), -1, dtype=np.int64)
    margin_q = np.full(n, np.inf, dtype=dt)
    if chunk is None:
        chunk = max(1, min(scene.n, (1 << 21) // max(n, 1)))
    for g0 in range(0, scene.n, chunk):
        g1 = min(scene.n, g0 + chunk)
        t1, _, w, tc = _pair_t(scene, origin, dirs, g0, g1)
        ok = (t1 > 0) & np.isfinite(t1)
        with np.errstate(invalid="ignore"):
            mq = np.where(tc > 0, np.abs(w), np.inf)
        margin_q = np.minimum(margin_q, np.min(mq, axis=1))
        t1m = np.where(ok, t1, np.inf)
        cat_t = np.concatenate([keep_t, t1m], axis=1)
        cat_w = np.concatenate([keep_w, w], axis=1)
        cat_i = np.concatenate([keep_i, np.broadcast_to(np.arange(g0, g1), t1.shape)], axis=1)
        order = np.argsort(cat_t, axis=1, kind="stable")[:, :K]
        keep_t = np.take_along_axis(cat_t, order, axis=1)
        keep_w = np.take_along_axis(cat_w, order, axis=1)
        keep_i = np.take_along_axis(cat_i, order, axis=1)
    sample = np.zeros((n, 3), dtype=dt)
    att = np.ones(n, dtype=dt)
    hits = np.zeros(n, dtype=np.int64)
    last_t = np.zeros(n, dtype=dt)
    ray_sh = None
    if scene.sh is not None:
        ray_sh = sh_basis(dirs / np.sqrt(np.sum(dirs * dirs, axis=-1, keepdims=True)), dt)
    prev_t = np.full(n, -1.0, dtype=dt)
    margin_c = np.full(n, np.inf, dtype=dt)
    dead = np.zeros(n, dtype=bool)
    for k in range(K):
        # an exact repeat of the previous entry distance is dropped: the reference's strict
        # `t1 > ray.start` (scene.py:433) cannot see it after ray_tracer.py:101
        valid = np.isfinite(keep_t[:, k]) & (keep_t[:, k] != prev_t) & (hits < depth)
        if t_cutoff > 0:
            margin_c = np.where(valid & ~dead, np.minimum(margin_c, np.abs(att - t_cutoff) / t_cutoff), margin_c)
            dead |= valid & (att < dt(t_cutoff))
            valid &= ~dead
        prev_t = np.where(np.isfinite(keep_t[:, k]), keep_t[:, k], prev_t)
        h = np.nonzero(valid)[0]
        if h.size == 0:
            continue
        gi = keep_i[h, k]
        alpha = scene.opacity[gi] * np.exp(keep_w[h, k] - dt(BOUNDING_THRESHOLD))
        rgb = scene.color[gi]
        if ray_sh is not None:
            rgb = rgb + np.einsum("rk,rkc->rc", ray_sh[h], scene.sh[gi])
        sample[h] += (att[h] * alpha)[:, None] * rgb
        att[h] *= dt(1) - alpha
        hits[h] += 1
        last_t[h] = keep_t[h, k]
    with np.errstate(invalid="ignore"):
        gaps = np.diff(keep_t, axis=1) / np.maximum(1.0, keep_t[:, 1:])
    gaps = np.where(np.isfinite(keep_t[:, 1:]), gaps, np.inf)
    margin_t = np.min(gaps, axis=1) if K > 1 else np.full(n, np.inf)
    start = np.where(hits == depth, last_t, np.inf).astype(dt)
    return dict(sample=sample, attenuation=att, start=start, hits=hits, idx=keep_i[:, :depth],
                t1=keep_t[:, :depth], margin_q=margin_q, margin_t=margin_t, margin_c=margin_c,
                fragile=(margin_q < eps_q) | (margin_t < eps_t) | (margin_c < eps_c))


def disp_buffer(sample, num_samples, num_steps, num_depth, dtype=np.float64):
    """ray_tracer.py:68-77 -- sample / (num_samples + num_steps/num_depth)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        return _a(sample, dtype) / dtype(dtype(num_samples) + dtype(num_steps) / dtype(num_depth))


# --------------------------------------------------------------------------------------
# scene.py:89-114 -- PLY ingest + activations
# --------------------------------------------------------------------------------------

def read_ply_vertices(path):
    """Minimal binary_little_endian PLY reader: returns {property: float32 array}."""
    with open(path, "rb") as f:
        data = f.read()
    end = data.index(b"end_header\n") + len(b"end_header\n")
    header = data[:end].decode("ascii").splitlines()
    assert header[0] == "ply" and "binary_little_endian" in header[1]
    n = 0
    props = []
    for line in header:
        tok = line.split()
        if tok[:2] == ["element", "vertex"]:
            n = int(tok[2])
        elif tok and tok[0] == "property":
            assert tok[1] == "float", "oracle reader handles float32 properties only"
            props.append(tok[2])
    arr = np.frombuffer(data, dtype="<f4", count=n * len(props), offset=end)
    arr = arr.reshape(n, len(props))
    return {p: arr[:, k].copy() for k, p in enumerate(props)}


def load_ply_gaussians(path, scale=1.0):
    """scene.py:95-114 -- wxyz -> xyzw, normalise quaternions, exp(scale)*scale, sigmoid of
    colour and opacity, f_rest reshaped (-1, 3, 15).  The kernel-side contract
    (scene.py:143-157: sh_field[i, k] is the RGB triple of coefficient k) is expressed here by
    transposing to (N, 15, 3); how Taichi 1.7.3 ingests the (N,3,15) array into an (N,15) vec3
    field cannot be verified here."""
    p = read_ply_vertices(path)
    pos = np.stack([p["x"], p["y"], p["z"]], axis=-1).astype(np.float64)
    rot = np.stack([p["rot_1"], p["rot_2"], p["rot_3"], p["rot_0"]], axis=-1).astype(np.float64)
    sc = np.stack([p["scale_0"], p["scale_1"], p["scale_2"]], axis=-1).astype(np.float64)
    col = np.stack([p["f_dc_0"], p["f_dc_1"], p["f_dc_2"]], axis=-1).astype(np.float64)
    sh = np.stack([p[f"f_rest_{k}"] for k in range(45)], axis=-1).astype(np.float64)
    sh = sh.reshape(-1, 3, 15).transpose(0, 2, 1)
    opa = p["opacity"].astype(np.float64)
    rot = rot / np.linalg.norm(rot, axis=-1)[:, None]
    sc = np.exp(sc) * scale
    col = 1 / (1 + np.exp(-col))
    opa = 1 / (1 + np.exp(-opa))
    return dict(position=pos, rotation=rot, scale=sc, color=col, opacity=opa, sh=sh)
