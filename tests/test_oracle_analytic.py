"""Hand-derived known answers for the parts of the path the reference's tests do not pin
(SURVEY.md 8c "golden vectors the build must create itself").  Each expectation is derived from
the cited reference lines, not from running the oracle."""

import math

import numpy as np
import pytest

from oracle import c_oracle
from oracle import rtgs_ref as ref

CAM = dict(position=(0.0, 0.0, 0.0), rotation=(0.0, 0.0, 0.0, 1.0), width=5, height=5, fx=5.0, fy=5.0)
CENTER = np.array([[2, 2]])  # pixel whose ray is exactly (0, 0, -1)
IDQ = (0.0, 0.0, 0.0, 1.0)


def render(arrays, cam=CAM, depth=16, pixels=CENTER, mode="topk", dtype=np.float64):
    osc = ref.OracleScene(**arrays, dtype=dtype)
    fn = ref.render_topk if mode == "topk" else ref.render_peel
    return fn(osc, cam["position"], cam["rotation"], cam["width"], cam["height"], cam["fx"], cam["fy"],
              depth=depth, pixels=pixels)


def scene(pos, scale, color, opacity, rot=None, sh=None):
    n = len(pos)
    return dict(position=np.array(pos, dtype=np.float64), rotation=np.array(rot or [IDQ] * n),
                scale=np.array(scale, dtype=np.float64), color=np.array(color, dtype=np.float64),
                opacity=np.array(opacity, dtype=np.float64), sh=sh)


def test_center_ray_direction_and_image_convention():
    # camera.py:31-55; docs coordinate-system.md: +x right, +y up, looks along -z, (0,0) bottom-left
    o, d = ref.generate_rays(CAM["position"], CAM["rotation"], 5, 5, 5.0, 5.0, pixels=[[2, 2], [4, 2], [2, 4], [0, 0]])
    assert d[0] == pytest.approx([0, 0, -1])
    assert d[1][0] > 0 and abs(d[1][1]) < 1e-15          # i grows to the right
    assert d[2][1] > 0 and abs(d[2][0]) < 1e-15          # j grows upwards
    assert d[3][0] < 0 and d[3][1] < 0                   # (0,0) is bottom-left
    # pxy = ((i+.5) - W/2)/f : pixel (4,2) -> (0.4, 0, -1)/|.|
    assert d[1] == pytest.approx(np.array([0.4, 0, -1]) / math.sqrt(1.16))
    assert np.linalg.norm(d, axis=1) == pytest.approx(np.ones(4))


@pytest.mark.parametrize("mode", ["topk", "peel"])
def test_isotropic_gaussian_on_axis(mode):
    # gaussian.py:215-226 with Sigma = s^2 I: t = D -/+ sqrt(3) s; chord midpoint = the centre, so
    # alpha = opacity (gaussian.py:197-198) and the pixel = opacity * color (ray_tracer.py:95).
    s, D = 0.2, 4.0
    r = render(scene([(0, 0, -D)], [(s, s, s)], [(0.2, 0.5, 0.9)], [0.7]), mode=mode)
    assert r["sample"][0] == pytest.approx(0.7 * np.array([0.2, 0.5, 0.9]), abs=1e-13)
    assert r["attenuation"][0] == pytest.approx(0.3, abs=1e-13)
    assert r["hits"][0] == 1 and math.isinf(r["start"][0])
    osc = ref.OracleScene(**scene([(0, 0, -D)], [(s, s, s)], [(1, 1, 1)], [1.0]))
    idx, t1, t2 = ref.closest_hit(osc, np.zeros(3), np.array([[0, 0, -1.0]]), 0.0, np.inf)
    assert idx[0] == 0 and t1[0] == pytest.approx(D - math.sqrt(3) * s) and t2[0] == pytest.approx(D + math.sqrt(3) * s)


def test_grazing_ray_alpha_tends_to_opacity_times_exp_minus_3():
    # at the silhouette delta -> 0, the midpoint sits on the Mahalanobis-3 shell: rho = e^-3
    s, D = 0.2, 4.0
    x = math.sqrt(3) * s * (1 - 1e-9)  # just inside the silhouette, ray along -z offset by x
    arrays = scene([(x, 0, -D)], [(s, s, s)], [(1, 1, 1)], [0.8])
    r = render(arrays)
    assert r["sample"][0, 0] == pytest.approx(0.8 * math.exp(-3), rel=1e-6)
    miss = render(scene([(math.sqrt(3) * s * (1 + 1e-9), 0, -D)], [(s, s, s)], [(1, 1, 1)], [0.8]))
    assert miss["sample"][0].tolist() == [0, 0, 0] and miss["hits"][0] == 0
    assert r["fragile"][0] and miss["fragile"][0]       # both flagged as next to a discontinuity


@pytest.mark.parametrize("mode", ["topk", "peel"])
def test_camera_inside_or_behind_is_invisible(mode):
    # t1 < 0 < t2 (inside) or t1 < t2 < 0 (behind) never satisfy ray.start < t1 (scene.py:433)
    inside = scene([(0, 0, -0.1)], [(1, 1, 1)], [(1, 1, 1)], [1.0])
    behind = scene([(0, 0, 4.0)], [(0.2, 0.2, 0.2)], [(1, 1, 1)], [1.0])
    for arrays in (inside, behind):
        r = render(arrays, mode=mode)
        assert r["sample"][0].tolist() == [0, 0, 0] and r["attenuation"][0] == 1 and r["hits"][0] == 0


@pytest.mark.parametrize("mode", ["topk", "peel"])
def test_two_stacked_gaussians_composite_front_to_back(mode):
    # ray_tracer.py:95-97: L = c1 a1 + (1 - a1) c2 a2, listed far-first to prove the sort
    c1, c2 = np.array([1.0, 0.0, 0.0]), np.array([0.0, 1.0, 0.0])
    arrays = scene([(0, 0, -6), (0, 0, -3)], [(0.3,) * 3, (0.1,) * 3], [c2, c1], [0.5, 0.6])
    r = render(arrays, mode=mode)
    assert r["sample"][0] == pytest.approx(0.6 * c1 + 0.4 * 0.5 * c2, abs=1e-13)
    assert r["attenuation"][0] == pytest.approx(0.4 * 0.5)


@pytest.mark.parametrize("mode", ["topk", "peel"])
def test_depth_truncation_ignores_the_17th(mode):
    # 17 coaxial Gaussians, depth 16 (__main__.py:70): the farthest is never composited
    n = 17
    pos = [(0, 0, -(2.0 + k)) for k in range(n)]
    col = [(1.0, 1.0, 1.0)] * n
    arrays = scene(pos, [(0.1,) * 3] * n, col, [0.1] * n)
    r = render(arrays, mode=mode)
    assert r["sample"][0, 0] == pytest.approx(1 - 0.9 ** 16, abs=1e-13)
    assert r["attenuation"][0] == pytest.approx(0.9 ** 16)
    assert r["hits"][0] == 16
    assert r["start"][0] == pytest.approx(2.0 + 15 - math.sqrt(3) * 0.1)
    assert render(arrays, depth=3, mode=mode)["sample"][0, 0] == pytest.approx(1 - 0.9 ** 3)


def test_exact_duplicates_collapse_to_one():
    # ray.start = t1 + 1e-8 is absorbed in fp32 and the strict t1 > start (scene.py:433) then
    # excludes an identical second Gaussian
    arrays = scene([(0, 0, -3), (0, 0, -3)], [(0.1,) * 3] * 2, [(1, 1, 1)] * 2, [0.5, 0.5])
    for mode in ("topk", "peel"):
        assert render(arrays, mode=mode)["sample"][0, 0] == pytest.approx(0.5)
    assert c_oracle.render(arrays, CAM, 16, CENTER)["sample"][0, 0] == pytest.approx(0.5)


def test_closed_form_alpha_equals_midpoint_evaluation():
    # gaussian.py:197 at x = o + (t1+t2)/2 d equals exp(delta/(4A) - 3)
    rng = np.random.default_rng(3)
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    minv = ref.gaussian_cov_inv(q, (0.3, 0.1, 0.2))
    mu = np.array([0.1, -0.05, -2.0])
    o, d = np.array([0.02, 0.01, 0.3]), np.array([0.03, -0.02, -1.0])
    d /= np.linalg.norm(d)
    t1, t2 = ref.gaussian_hit(o, d, mu, minv)
    assert np.isfinite(t1)
    _, alpha = ref.gaussian_eval(o + (t1 + t2) / 2 * d, d, mu, minv, (1, 1, 1), 0.9)
    m = o - mu
    A, B, C = d @ minv @ d, 2 * d @ minv @ m, m @ minv @ m - 3
    assert alpha == pytest.approx(0.9 * math.exp((B * B - 4 * A * C) / (4 * A) - 3), rel=1e-12)


def test_sh_basis_including_the_y33_quirk():
    # gaussian.py:149-163; y_33 = 1/4 c6 (5 z^2 - 3 z) as coded (:160), NOT 5 z^3 - 3 z
    d = np.array([0.36, 0.48, 0.8])
    Y = ref.sh_basis(d)
    c6 = math.sqrt(7 / math.pi)
    assert Y[11] == pytest.approx(0.25 * c6 * (5 * 0.64 - 3 * 0.8))
    assert Y[11] != pytest.approx(0.25 * c6 * (5 * 0.512 - 3 * 0.8))
    assert Y[0] == pytest.approx(0.5 * math.sqrt(3 / math.pi) * 0.48)
    assert Y[9] == pytest.approx(0.5 * math.sqrt(105 / math.pi) * 0.36 * 0.48 * 0.8)
    # colour = color + sum_k Y_k sh_k (gaussian.py:199-200), direction normalised first
    sh = np.zeros((1, 15, 3))
    sh[0, 11] = (1.0, 2.0, 3.0)
    rgb, _ = ref.gaussian_eval(np.zeros(3), 5 * d, np.zeros(3), np.eye(3), (0.1, 0.1, 0.1), 1.0, sh[0])
    assert rgb == pytest.approx(0.1 + Y[11] * np.array([1.0, 2.0, 3.0]))


def test_sh_colour_reaches_the_pixel():
    sh = np.zeros((1, 15, 3))
    sh[0, 1] = (0.5, 0.0, 0.0)     # sh_11 multiplies 1/2 c0 z; the centre ray has z = -1
    arrays = scene([(0, 0, -4)], [(0.2,) * 3], [(0.5, 0.5, 0.5)], [1.0], sh=sh)
    want = 0.5 + 0.5 * math.sqrt(3 / math.pi) * (-1) * 0.5
    for mode in ("topk", "peel"):
        assert render(arrays, mode=mode)["sample"][0] == pytest.approx([want, 0.5, 0.5], abs=1e-13)
    assert c_oracle.render(arrays, CAM, 16, CENTER)["sample"][0] == pytest.approx([want, 0.5, 0.5], abs=1e-13)


def test_disp_buffer():
    # ray_tracer.py:68-77
    s = np.full((2, 2, 3), 6.0)
    assert ref.disp_buffer(s, 1, 8, 16) == pytest.approx(s / 1.5)
    assert ref.disp_buffer(s, 2, 0, 16) == pytest.approx(s / 2)
