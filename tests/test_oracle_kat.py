"""Pins the oracle against every value the REFERENCE's own tests hold for this path
(SURVEY.md 8c): tests/test_quaternion.py, tests/test_ray.py, tests/test_gaussian.py and the
tests/data/test.ply fixture.  Expectations the reference states but its code contradicts
(test_quaternion.py:58-62 `inv`, test_gaussian.py:83-84 `hit == (0, inf)`) are asserted at the
value the reference's ARITHMETIC gives, and say so."""

import math
import os

import numpy as np
import pytest

from oracle import rtgs_ref as ref

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_mul_known_answer(dtype):
    # reference tests/test_quaternion.py:25-29
    pq = ref.quat_mul((1, -2, 1, 3), (-1, 2, 3, 2), dtype)
    assert pq.tolist() == [-9, -2, 11, 8]


def test_conj_known_answer():
    # reference tests/test_quaternion.py:40-43
    assert ref.quat_conj((1, 2, 3, 4)).tolist() == [-1, -2, -3, 4]


def test_inv_is_literal():
    # reference tests/test_quaternion.py:58-62 expects q*inv(q) == (0,0,0,1) for q=(1,2,3,4); the
    # code (utils/quaternion.py:47) divides by |q|, not |q|^2, which gives (0,0,0,sqrt(30)).
    q = np.array([1.0, 2.0, 3.0, 4.0])
    assert ref.quat_mul(q, ref.quat_inv(q)) == pytest.approx([0, 0, 0, math.sqrt(30)], abs=1e-12)
    unit = q / np.linalg.norm(q)
    assert ref.quat_mul(unit, ref.quat_inv(unit)) == pytest.approx([0, 0, 0, 1], abs=1e-12)


def test_axis_angle_round_trip():
    # reference tests/test_quaternion.py:81-90
    v = np.array([0.0, math.pi, 0.0])
    q = ref.quat_from_axis_angle(v, np.float32)
    assert q == pytest.approx([0, 1, 0, 0], abs=1e-6)
    assert ref.quat_as_axis_angle(q, np.float32) == pytest.approx(v, abs=1e-6)
    assert ref.quat_from_axis_angle(np.zeros(3)).tolist() == [0, 0, 0, 1]


def test_rot_vec3_known_answers():
    # reference tests/test_quaternion.py:107-118
    qz = ref.quat_from_axis_angle((0, 0, math.pi / 2), np.float32)
    assert ref.quat_rot_vec3(qz, (1, 0, 0), np.float32) == pytest.approx([0, 1, 0], abs=1e-6)
    qy = ref.quat_from_axis_angle((0, math.pi / 2, 0), np.float32)
    assert ref.quat_rot_vec3(qy, (1, 0, 0), np.float32) == pytest.approx([0, 0, -1], abs=1e-6)


@pytest.mark.parametrize("seed", range(5))
def test_rotation_matrices_match_rot_vec3(seed):
    # reference tests/test_quaternion.py:147-153, 182-188
    rng = np.random.default_rng(seed)
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    v = rng.normal(size=3)
    want = ref.quat_rot_vec3(q, v, np.float32)
    assert ref.quat_as_rotation_mat3(q, np.float32) @ v.astype(np.float32) == pytest.approx(want, abs=1e-6)
    m4 = ref.quat_as_rotation_mat4(q, np.float32)
    assert (m4 @ np.append(v, 1).astype(np.float32))[:3] == pytest.approx(want, abs=1e-6)
    assert m4[3].tolist() == [0, 0, 0, 1]


def test_ray_get_and_defaults():
    # reference tests/test_ray.py:15-28, 53-116
    assert ref.RAY_DEFAULT == dict(origin=(0, 0, 0), direction=(0, 1, 0), start=0, end=math.inf)
    rng = np.random.default_rng(42)
    o, d, t = rng.random((4, 4, 3)), rng.random((4, 4, 3)), rng.random((4, 4))
    got = ref.ray_get(o, d, t, np.float32)
    assert got == pytest.approx((o + t[..., None] * d), abs=1e-6)


def test_gaussian_defaults_and_hit_of_the_default_gaussian():
    # reference tests/test_gaussian.py:19-24: defaults.
    g = ref.GAUSSIAN_DEFAULT
    assert g == dict(position=(0, 0, 0), rotation=(0, 0, 0, 1), scale=(1, 1, 1), color=(1, 0, 1), opacity=1)
    # tests/test_gaussian.py:65-84 expects (0, inf) for a ray from the centre of the unit Gaussian;
    # gaussian.py:215-226 evaluates to A=1, B=0, C=-3 => (-sqrt3, +sqrt3).
    minv = ref.gaussian_cov_inv(g["rotation"], g["scale"])
    assert minv == pytest.approx(np.eye(3), abs=1e-15)
    t1, t2 = ref.gaussian_hit((0, 0, 0), (0, 1, 0), g["position"], minv)
    assert (float(t1), float(t2)) == pytest.approx((-math.sqrt(3), math.sqrt(3)), abs=1e-12)


def test_ply_fixture_decodes_as_surveyed():
    # reference tests/test_scene.py:16-23 loads tests/data/test.ply; SURVEY.md 8c (8) lists its
    # content: 16 vertices x 62 float32 properties, header 1527 B + 3968 B payload.
    path = os.path.join(GOLDEN, "test.ply")
    assert os.path.getsize(path) == 1527 + 3968
    props = ref.read_ply_vertices(path)
    assert len(props) == 62 and all(v.shape == (16,) for v in props.values())
    g = ref.load_ply_gaussians(path)
    assert g["position"].shape == (16, 3) and g["sh"].shape == (16, 15, 3)
    assert g["position"][:, 0].min() >= -0.68 - 1e-2 and g["position"][:, 0].max() <= 1.35 + 1e-2
    assert g["position"][:, 2].min() >= -4.64 - 1e-2 and g["position"][:, 2].max() <= 1.34 + 1e-2
    assert 4.0e-4 <= g["scale"].min() and g["scale"].max() <= 7.1e-2
    assert np.linalg.norm(g["rotation"], axis=1) == pytest.approx(np.ones(16), abs=1e-12)
    assert np.abs(g["sh"]).max() <= 0.045
    raw_opacity = props["opacity"]
    assert -2.83 <= raw_opacity.min() and raw_opacity.max() <= 3.12
    assert g["opacity"] == pytest.approx(1 / (1 + np.exp(-raw_opacity.astype(np.float64))))
    # scale argument multiplies exp(scale) (scene.py:112)
    assert ref.load_ply_gaussians(path, 2.0)["scale"] == pytest.approx(2 * g["scale"])
