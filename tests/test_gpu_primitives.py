"""-m gpu: the device mirrors of the reference's primitives, checked against the reference's own
known answers (tests/test_quaternion.py, test_ray.py, test_gaussian.py, test_scene.py) and
against the oracle on seeded inputs."""

import math
import os

import numpy as np
import pytest
import torch

import gpu_util as U
import rtgs
import rtgs.utils.quaternion as quat
from oracle import rtgs_ref as ref
from rtgs import _native as nat
from rtgs.gaussian import Gaussian, new_gaussian
from rtgs.ray import Ray, new_ray
from rtgs.utils.types import vec3, vec4

pytestmark = pytest.mark.gpu


def np_(t):
    return t.detach().cpu().numpy()


# ---- quaternion (reference tests/test_quaternion.py) ----

def test_quat_known_answers():
    assert np_(quat.mul((1, -2, 1, 3), (-1, 2, 3, 2))).tolist() == [-9, -2, 11, 8]      # :25-29
    assert np_(quat.conj((1, 2, 3, 4))).tolist() == [-1, -2, -3, 4]                      # :40-43
    q = (1.0, 2.0, 3.0, 4.0)                                                             # :58-62, literal
    assert np_(quat.mul(q, quat.inv(q))) == pytest.approx([0, 0, 0, math.sqrt(30)], abs=1e-5)
    v = (0.0, math.pi, 0.0)                                                              # :81-90
    qa = quat.from_axis_angle(v)
    assert np_(qa) == pytest.approx([0, 1, 0, 0], abs=1e-6)
    assert np_(quat.as_axis_angle(qa)) == pytest.approx(v, abs=1e-6)
    qz = quat.from_axis_angle((0, 0, math.pi / 2))                                       # :107-118
    assert np_(quat.rot_vec3(qz, (1, 0, 0))) == pytest.approx([0, 1, 0], abs=1e-6)
    qy = quat.from_axis_angle((0, math.pi / 2, 0))
    assert np_(quat.rot_vec3(qy, (1, 0, 0))) == pytest.approx([0, 0, -1], abs=1e-6)


def test_rotation_matrices_match_rot_vec3_and_oracle():
    rng = np.random.default_rng(42)                                                       # :147-153, 182-188
    q = rng.normal(size=(64, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    v = rng.normal(size=(64, 3)).astype(np.float32)
    m3, m4, rv = np_(quat.as_rotation_mat3(q)), np_(quat.as_rotation_mat4(q)), np_(quat.rot_vec3(q, v))
    assert np.einsum("nij,nj->ni", m3, v) == pytest.approx(rv, abs=1e-6)
    assert np.einsum("nij,nj->ni", m4[:, :3, :3], v) == pytest.approx(rv, abs=1e-6)
    assert m3 == pytest.approx(ref.quat_as_rotation_mat3(q.astype(np.float64)), abs=1e-6)
    p = rng.normal(size=(64, 4)).astype(np.float32)
    assert np_(quat.mul(p, q)) == pytest.approx(ref.quat_mul(p.astype(np.float64), q.astype(np.float64)), abs=1e-5)


# ---- ray (reference tests/test_ray.py) ----

def test_ray_field_and_get():
    f = Ray.field(shape=(4, 4))                                                          # :31-50
    assert f.shape == (4, 4)
    r = f[0, 0]
    assert r.origin == vec3(0) and r.direction == vec3(0) and r.start == 0 and r.end == 0
    f.init()
    r = f[3, 2]
    assert r.origin == vec3(0) and r.direction == vec3(0, 1, 0) and r.start == 0 and math.isinf(r.end)
    rng = np.random.default_rng(42)                                                      # :53-116
    f.data[..., :6] = torch.from_numpy(rng.random((4, 4, 6)).astype(np.float32)).cuda()
    t = rng.random((4, 4)).astype(np.float32)
    got = np_(f.get(t))
    o, d = np_(f.origin), np_(f.direction)
    assert got == pytest.approx(o + t[..., None] * d, abs=1e-6)
    one = f[1, 2]
    assert one.get(0.5) == vec3(got[1, 2] * 0 + (one.origin + np.float32(0.5) * one.direction))
    f[0, 1] = new_ray(vec3(1, 2, 3), vec3(0, 0, 1), 1, 2)
    assert f[0, 1].origin == vec3(1, 2, 3) and f[0, 1].end == 2


# ---- gaussian (reference tests/test_gaussian.py) ----

def test_gaussian_field_defaults():
    f = Gaussian.field(shape=(32,))                                                      # :42-63
    assert f.shape == (32,)
    g = f[0]
    assert g.position == vec3(0) and g.rotation == vec4(0) and g.scale == vec3(0) and g.opacity == 0
    f.init()
    g = f[31]
    assert g.rotation == vec4(0, 0, 0, 1) and g.scale == vec3(1, 1, 1) and g.color == vec3(1, 0, 1) and g.opacity == 1
    f[5] = new_gaussian(vec3(1, 2, 3), vec4(0, 0, 0, 1), vec3(2, 3, 4), vec3(1, 0, 0), 0.75)
    assert f[5].position == vec3(1, 2, 3) and f[5].opacity == 0.75


def test_gaussian_hit_of_the_default_gaussian():
    # tests/test_gaussian.py:65-84 states (0, inf); gaussian.py:215-226 evaluates to -/+ sqrt(3)
    g = new_gaussian()
    h = g.hit(new_ray())
    assert (h.x, h.y) == pytest.approx((-math.sqrt(3), math.sqrt(3)), abs=1e-6)
    h = g.hit(new_ray(vec3(0, 0, 0), vec3(0, 1, 0), 0, 1))
    assert (h.x, h.y) == pytest.approx((-math.sqrt(3), math.sqrt(3)), abs=1e-6)
    miss = g.hit(new_ray(vec3(5, 0, 0), vec3(0, 1, 0)))
    assert math.isinf(miss.x) and math.isinf(miss.y)


def test_bounding_box_is_the_box_of_the_six_axis_end_points():
    # gaussian.py:104-138: min / max over position +- R S (3 e_k) -- analytic for the identity rotation (+- 3 s), a
    # quarter turn about z (x and y extents swap), and against the oracle's restatement for random Gaussians
    g = new_gaussian(vec3(1, 2, 3), vec4(0, 0, 0, 1), vec3(0.5, 1.0, 2.0))
    b = g.bounding_box()
    assert np.allclose(b.p_min, (1 - 1.5, 2 - 3.0, 3 - 6.0), atol=1e-6) and np.allclose(b.p_max, (2.5, 5.0, 9.0), atol=1e-6)
    assert np.allclose(b.get_centroid(), (1, 2, 3), atol=1e-6) and np.allclose(b.size(), (3, 6, 12), atol=1e-6)
    s2 = math.sqrt(0.5)
    b = new_gaussian(vec3(0), vec4(0, 0, s2, s2), vec3(0.5, 1.0, 2.0)).bounding_box()
    assert np.allclose(b.p_max, (3.0, 1.5, 6.0), atol=1e-5) and np.allclose(b.p_min, (-3.0, -1.5, -6.0), atol=1e-5)
    rng = np.random.default_rng(11)
    for _ in range(8):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        pos, sc = rng.normal(size=3), rng.uniform(0.05, 2.0, 3)
        b = new_gaussian(pos, q, sc).bounding_box()
        lo, hi = ref.gaussian_bounding_box(np.float32(pos), np.float32(q), np.float32(sc))
        assert np.allclose(b.p_min, lo, atol=2e-6 * (1 + np.abs(lo).max())) and np.allclose(b.p_max, hi, atol=2e-6 * (1 + np.abs(hi).max()))


def test_cov_hit_eval_match_the_oracle():
    rng = np.random.default_rng(7)
    for _ in range(8):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        g = new_gaussian(rng.normal(size=3), q, rng.uniform(0.2, 1.5, 3), rng.random(3), rng.uniform(0.1, 0.9))
        for k, name in enumerate(rtgs.gaussian.SH_NAMES):
            setattr(g, name, vec3(rng.uniform(-0.3, 0.3, 3)))
        pos64, rot64, sc64 = (np.asarray(a, np.float64) for a in (g.position, g.rotation, g.scale))
        assert g.cov() == pytest.approx(ref.gaussian_cov(rot64, sc64), rel=2e-6, abs=1e-7)
        minv = ref.gaussian_cov_inv(rot64, sc64)
        o, d = rng.normal(size=3).astype(np.float32), rng.normal(size=3).astype(np.float32)
        t1, t2 = ref.gaussian_hit(o.astype(np.float64), d.astype(np.float64), pos64, minv)
        h = g.hit(new_ray(o, d))
        if np.isfinite(t1):
            assert (h.x, h.y) == pytest.approx((float(t1), float(t2)), rel=1e-5, abs=1e-5)
        else:
            assert math.isinf(h.x)
        x = (pos64 + rng.normal(size=3) * 0.5).astype(np.float32)
        rgb, alpha = ref.gaussian_eval(x.astype(np.float64), d.astype(np.float64), pos64, minv,
                                       np.asarray(g.color, np.float64), g.opacity, g.sh_array().astype(np.float64))
        e = g.eval(x, d)
        assert np.asarray(e)[:3] == pytest.approx(rgb, abs=2e-6) and e.w == pytest.approx(alpha, rel=2e-5, abs=1e-7)
        dn = d / np.linalg.norm(d)
        assert np.asarray(g.eval_sh(dn)) == pytest.approx(
            ref.gaussian_eval_sh(dn.astype(np.float64), g.sh_array().astype(np.float64)), abs=2e-6)


# ---- camera ----

def test_generate_ray_field_matches_the_oracle():
    arrays, cam, _ = U.case("oblique")
    camera = rtgs.Camera(cam["position"], cam["rotation"], (cam["width"], cam["height"]), (cam["fx"], cam["fy"]))
    camera.generate_ray_field(camera.position, camera.rotation)
    rays = np_(camera.cam_ray_field.data)
    _, d = ref.generate_rays(cam["position"], cam["rotation"], cam["width"], cam["height"], cam["fx"], cam["fy"])
    assert rays[..., 3:6].reshape(-1, 3) == pytest.approx(d, abs=3e-7)
    assert (rays[..., 0:3] == np.asarray(cam["position"], np.float32)).all()
    assert (rays[..., 6] == 0).all() and np.isinf(rays[..., 7]).all()
    one = camera.generate_ray(camera.position, camera.rotation, (0.5, 0.25))
    _, d1 = ref.generate_rays(cam["position"], cam["rotation"], 2, 4, cam["fx"] * 2 / cam["width"],
                              cam["fy"] * 4 / cam["height"], pixels=[[0, 0]])
    # uv (0.5, 0.25) of the (150, 101) sensor: pxy = ((0.5-0.5)*150/fx, (0.25-0.5)*101/fy)
    want = np.array([0.0, -0.25 * cam["height"] / cam["fy"], -1.0])
    want = ref.quat_rot_vec3(np.asarray(cam["rotation"]), want / np.linalg.norm(want))
    assert np.asarray(one.direction) == pytest.approx(want, abs=3e-7)


# ---- scene ----

def test_scene_hit_matches_the_oracle():
    arrays, cam, _ = U.case("c1")
    scene = rtgs.Scene.from_arrays(**arrays)
    osc = ref.OracleScene(**arrays)
    rng = np.random.default_rng(13)
    n = 3000
    o = rng.uniform(-1.5, 1.5, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    start = rng.uniform(0, 0.5, n).astype(np.float32)
    end = np.where(rng.random(n) < 0.3, rng.uniform(0.5, 3.0, n), np.inf).astype(np.float32)
    rays = np.concatenate([o, d, start[:, None], end[:, None]], axis=1).astype(np.float32)
    idx, t = scene.hit_batch(torch.from_numpy(rays).cuda())
    wi, wt1, wt2 = ref.closest_hit(osc, o.astype(np.float64), d.astype(np.float64), start.astype(np.float64),
                                   end.astype(np.float64))
    idx, t = np_(idx), np_(t)
    assert (idx == wi).mean() >= 0.999          # fp32 rounding of the returned t may flip a near-tie
    same = idx == wi
    hit = same & (wi >= 0)
    assert hit.sum() > 500 and (wi < 0).sum() > 50
    assert t[hit, 0] == pytest.approx(wt1[hit], rel=1e-6, abs=1e-6)
    assert t[hit, 1] == pytest.approx(wt2[hit], rel=1e-6, abs=1e-6)
    assert np.isinf(t[same & (wi < 0)]).all()
    k = int(np.nonzero(hit)[0][0])
    single = scene.hit(Ray.from_array(rays[k]))
    assert single.gaussian_idx == wi[k] and single.intersections.x == pytest.approx(wt1[k], rel=1e-6)
    assert scene.hit(new_ray(vec3(50, 50, 50), vec3(0, 1, 0))).gaussian_idx == -1


def test_ply_sh_layouts_and_crlf_header(tmp_path):
    # f_rest of an INRIA file is channel-major (15 R, 15 G, 15 B); the reference reshapes it
    # (-1, 3, 15) for a (N, 15) vec3 field (scene.py:106-107, 122, 127).  Both readings are offered
    # and each is pinned by a hand-computed pixel: one Gaussian on the optical axis, viewed along
    # d = (0, 0, -1), where only Y_11, Y_22 and Y_33 are non-zero (gaussian.py:149-163).
    import math
    names = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(45)] + \
            ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    f_rest = (np.arange(45, dtype=np.float32) - 22.0) * np.float32(0.004)
    row = np.concatenate([[0, 0, -4, 0, 0, 0, 0.2, -0.1, 0.4], f_rest, [50.0, math.log(0.2)] * 1 + [math.log(0.2)] * 2,
                          [1, 0, 0, 0]]).astype("<f4")
    for eol in (b"\n", b"\r\n"):
        header = eol.join([b"ply", b"format binary_little_endian 1.0", b"element vertex 1"] +
                          [b"property float " + n.encode() for n in names] + [b"end_header"]) + eol
        path = tmp_path / ("crlf.ply" if eol == b"\r\n" else "lf.ply")
        path.write_bytes(header + row.tobytes())
        c0, c2, c6 = math.sqrt(3 / math.pi), math.sqrt(5 / math.pi), math.sqrt(7 / math.pi)
        Y = {1: 0.5 * c0 * -1.0, 5: 0.25 * c2 * 2.0, 11: 0.25 * c6 * (5.0 + 3.0)}     # y_33 as coded: 5 z^2 - 3 z
        base = 1.0 / (1.0 + np.exp(-np.array([0.2, -0.1, 0.4])))
        for layout, coef in (("inria", lambda k, c: f_rest[c * 15 + k]), ("reference_raw", lambda k, c: f_rest[3 * k + c])):
            scene = rtgs.Scene()
            scene.load_file(path, sh_layout=layout)
            camera = rtgs.Camera((0, 0, 0), (0, 0, 0, 1), (5, 5), (5.0, 5.0))
            tracer = rtgs.RayTracer((5, 5), scene, camera)
            tracer.sample_all(16)
            torch.cuda.synchronize()
            want = [base[c] + sum(Y[k] * coef(k, c) for k in Y) for c in range(3)]    # opacity sigmoid(50) = 1
            assert np_(tracer.sample_buf[2, 2]) == pytest.approx(want, abs=3e-6), (layout, eol)
    with pytest.raises(ValueError):
        rtgs.Scene().load_file(path, sh_layout="transposed")


def test_ply_ingest_at_capture_scale_renders_like_its_arrays(tmp_path):
    # SURVEY 8(f)-2 at the size of the reference's use case (truck.ply, ~150 k Gaussians,
    # docs/source/get-started.md:64-74): an INRIA-layout binary PLY (raw log-scales, logit opacities,
    # wxyz quaternions, f_dc, channel-major f_rest) written here, loaded through Scene.load_file
    # (scene.py:95-114: activations, normalisation, wxyz -> xyzw) and rendered; must equal the scene
    # built directly from the post-activation arrays, and the C oracle on sampled pixels
    from oracle import c_oracle
    from rtgs import synthetic
    n, w, h = 60_000, 240, 135
    arrays = synthetic.make_capture_scene(n, w / h, 99)
    rng = np.random.default_rng(8)
    sh = (rng.random((n, 15, 3)) * 0.1 - 0.05).astype(np.float32)
    q = arrays["rotation"] * rng.uniform(0.5, 2.0, (n, 1)).astype(np.float32)      # the loader normalises
    raw = np.concatenate([
        arrays["position"], np.zeros((n, 3), np.float32),
        np.log(arrays["color"] / (1 - arrays["color"])),                              # f_dc: sigmoid^-1
        sh.transpose(0, 2, 1).reshape(n, 45),                                         # channel-major f_rest
        np.log(arrays["opacity"] / (1 - arrays["opacity"]))[:, None],
        np.log(arrays["scale"] / np.float32(1.5)),                                    # load_file(scale=1.5)
        q[:, [3, 0, 1, 2]]], axis=1).astype("<f4")
    names = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(45)] + \
            ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    header = "\n".join(["ply", "format binary_little_endian 1.0", f"element vertex {n}"] +
                       [f"property float {nm}" for nm in names] + ["end_header"]) + "\n"
    path = tmp_path / "capture.ply"
    path.write_bytes(header.encode() + raw.tobytes())
    scene = rtgs.Scene()
    scene.load_file(path, 1.5)
    f = scene.gaussian_field
    assert f.n == n
    for name, want in (("position", arrays["position"]), ("rotation", arrays["rotation"]), ("scale", arrays["scale"]),
                       ("color", arrays["color"]), ("opacity", arrays["opacity"]), ("sh", sh)):
        assert np_(getattr(f, name)) == pytest.approx(want, rel=3e-5, abs=3e-6), name
    cam = synthetic.bench_camera(w, h)
    camera = rtgs.Camera(cam["position"], cam["rotation"], (w, h), (cam["fx"], cam["fy"]))
    tracer = rtgs.RayTracer((w, h), scene, camera)
    tracer.sample_all(16)
    torch.cuda.synchronize()
    loaded = {k: np_(getattr(f, k)) for k in ("position", "rotation", "scale", "color", "opacity", "sh")}
    pix = np.stack([rng.integers(0, w, 1500), rng.integers(0, h, 1500)], axis=-1).astype(np.int32)
    want = c_oracle.render(loaded, cam, 16, pix)
    U.assert_parity(U.at(tracer.sample_buf, pix), want, "PLY scene at capture scale")
    assert want["hits"].mean() > 4


def test_accelerated_scene_hit_equals_brute_force_bit_for_bit():
    # the slab-box structure (what the reference's BVH is to Scene.hit, scene.py:162-450) only culls
    # Gaussians a ray cannot hit: indices AND distances must equal the brute-force kernel's exactly,
    # for coherent camera rays and for hostile ones (axis-parallel and zero directions, origins on box
    # planes, unnormalised directions, finite windows), in Morton order and in the caller's order
    from rtgs import synthetic
    arrays = synthetic.make_capture_scene(20_000, 1.5, 77)
    cam = synthetic.bench_camera(96, 64)
    rng = np.random.default_rng(3)
    for spatial in (True, False):
        scene = rtgs.Scene()
        scene.sort_spatially = spatial
        scene.set_arrays(**arrays)
        camera = rtgs.Camera(cam["position"], cam["rotation"], (96, 64), (cam["fx"], cam["fy"]))
        camera.generate_ray_field(camera.position, camera.rotation)
        coherent = camera.cam_ray_field.data.reshape(-1, 8).clone()
        n = 6000
        o = rng.uniform(-1.6, 1.6, (n, 3)).astype(np.float32)
        d = rng.normal(size=(n, 3)).astype(np.float32) * rng.uniform(0.2, 3.0, (n, 1)).astype(np.float32)
        d[:300, 0] = 0.0; d[300:600, 1:] = 0.0; d[600:610] = 0.0
        o[:300, 0] = arrays["position"][:300, 0]          # origins on planes through Gaussian centres
        start = np.where(rng.random(n) < 0.5, 0.0, rng.uniform(-0.5, 1.0, n)).astype(np.float32)
        end = np.where(rng.random(n) < 0.4, rng.uniform(0.3, 3.0, n), np.inf).astype(np.float32)
        hostile = torch.from_numpy(np.concatenate([o, d, start[:, None], end[:, None]], axis=1)).cuda()
        for rays in (coherent, hostile):
            ia, ta = scene.hit_batch(rays)
            ib, tb = scene.hit_batch(rays, brute_force=True)
            torch.cuda.synchronize()
            assert torch.equal(ia, ib) and torch.equal(ta, tb)
            assert int((ia >= 0).sum()) > 500
    assert scene._accel is not None and scene._accel[0] == scene.geometry_key()
    v = scene.version
    scene.commit()                                         # a re-committed scene rebuilds its structure
    scene.hit_batch(coherent)
    assert scene._accel[0][0] == v + 1
    assert nat.lib().rtgs_closest_hit_accel(scene.native(), nat.ptr(scene.accel()), 64, 10, nat.ptr(coherent),
                                            nat.ptr(ia), nat.ptr(ta), nat.stream()) == -3


def test_accelerated_scene_hit_orders_do_not_change_ties_or_answers():
    # nearest-first traversal (runs of 64 groups, groups inside a run) and the bounding-ball reject:
    # 80 k Gaussians = 79 groups = 2 runs, the last 3 000 exact copies of the first 3 000 -- every hit
    # of a copied Gaussian is an exact t1 tie between two indices that sit in different runs (caller's
    # order) or side by side (Morton order); brute force keeps the lower index, and so must this
    from rtgs import synthetic
    arrays = synthetic.make_scene(80_000, 1.5, 91)
    for k in ("position", "rotation", "scale", "color", "opacity"):
        arrays[k][-3000:] = arrays[k][:3000]
    arrays["scale"][:3000] *= 2.0
    arrays["scale"][-3000:] *= 2.0          # big enough to be hit often
    rng = np.random.default_rng(5)
    n = 8000
    o = rng.uniform(-2.5, 2.5, (n, 3)).astype(np.float32)
    tgt = arrays["position"][rng.integers(0, 3000, n)] + rng.normal(size=(n, 3)).astype(np.float32) * 0.01
    d = (tgt - o).astype(np.float32)
    d[n // 2:] = rng.normal(size=(n - n // 2, 3)).astype(np.float32)
    rays = torch.from_numpy(np.concatenate([o, d, np.zeros((n, 1), np.float32), np.full((n, 1), np.inf, np.float32)], axis=1)).cuda()
    ties = 0
    for spatial in (False, True):
        scene = rtgs.Scene()
        scene.sort_spatially = spatial
        scene.set_arrays(**arrays)
        ia, ta = scene.hit_batch(rays)
        ib, tb = scene.hit_batch(rays, brute_force=True)
        torch.cuda.synchronize()
        assert torch.equal(ia, ib) and torch.equal(ta, tb)
        if not spatial:      # caller's order: indices are the caller's, a copied Gaussian must answer as its original
            hit = ia[ia >= 0]
            ties = int((hit < 3000).sum())
            assert int((hit >= 77_000).sum()) == 0
    assert ties > 500


def test_scene_load_file():
    # reference tests/test_scene.py:16-23: loading tests/data/test.ply must not raise
    path = os.path.join(U.GOLDEN, "test.ply")
    scene = rtgs.Scene()
    scene.load_file(__import__("pathlib").Path(path))
    want = ref.load_ply_gaussians(path)
    f = scene.gaussian_field
    assert f.n == 16
    for name in ("position", "rotation", "scale", "color", "opacity", "sh"):
        assert np_(getattr(f, name)) == pytest.approx(want[name], rel=2e-6, abs=1e-7), name
    # covariance kernel on the loaded Gaussians vs the oracle (scales down to 4e-4 -> 6e6 entries)
    minv = ref.gaussian_cov_inv(np_(f.rotation).astype(np.float64), np_(f.scale).astype(np.float64))
    cinv = torch.empty((16, 6), dtype=torch.float32, device="cuda")
    nat.check(nat.lib().rtgs_covariance(16, nat.ptr(f.rotation), nat.ptr(f.scale), None, nat.ptr(cinv), nat.stream()))
    c = np_(cinv)
    got = np.stack([c[:, [0, 1, 2]], c[:, [1, 3, 4]], c[:, [2, 4, 5]]], axis=1)
    assert got == pytest.approx(minv, rel=5e-6)
    scene2 = rtgs.Scene()
    scene2.load_file(__import__("pathlib").Path(path), 2.0)
    assert np_(scene2.gaussian_field.scale) == pytest.approx(2 * want["scale"], rel=2e-6)


def test_ply_scene_renders_like_the_oracle():
    from oracle import c_oracle
    path = os.path.join(U.GOLDEN, "test.ply")
    scene = rtgs.Scene()
    scene.load_file(__import__("pathlib").Path(path), 8.0)      # --scale 8 makes the 16 splats visible
    f = scene.gaussian_field
    arrays = {k: np_(getattr(f, k)) for k in ("position", "rotation", "scale", "color", "opacity", "sh")}
    cam = dict(position=(0.3, 0.0, 3.0), rotation=(0.0, 0.0, 0.0, 1.0), width=96, height=64, fx=60.0, fy=60.0)
    cam["position"] = tuple(float(np.float32(v)) for v in cam["position"])
    camera = rtgs.Camera(cam["position"], cam["rotation"], (96, 64), (60.0, 60.0))
    tracer = rtgs.RayTracer((96, 64), scene, camera)
    tracer.sample_all(16)
    want = c_oracle.render(arrays, cam, 16, eps_q=1e-4)
    assert np.abs(want["sample"]).max() > 0.05
    U.assert_parity(np_(tracer.sample_buf).reshape(-1, 3), want, "test.ply")


def test_disp_and_fill():
    arrays, cam, depth = U.case("c1")
    _, _, tr = U.build(arrays, cam)
    tr.sample_buf.fill_(6.0)
    tr.generate_disp_buffer(1, 8, 16)
    assert float((tr.disp_buf - 4.0).abs().max()) == 0.0
    tr.clear_attenuation()
    assert float(tr.attenuation_buf.min()) == 1.0 == float(tr.attenuation_buf.max())


def test_cli_renders_headless(tmp_path):
    # reference CLI flags (__main__.py:34-85) on the bundled fixture and on a synthetic scene
    from rtgs.__main__ import main
    out = tmp_path / "ply.ppm"
    assert main(["-o", os.path.join(U.GOLDEN, "test.ply"), "-r", "96,64", "-f", "60", "-d", "4", "-s", "2",
                 "--scale", "8", "--radius", "3", "--out", str(out)]) == 0
    assert out.stat().st_size == len(b"P6\n96 64\n255\n") + 96 * 64 * 3
    out2 = tmp_path / "syn.png"
    assert main(["--synthetic", "2000", "-r", "128,72", "--radius", "3", "--one-pass", "--out", str(out2)]) == 0
    assert out2.read_bytes().startswith(b"\x89PNG")


def test_cli_image_matches_the_oracle_at_an_orbit_pose(tmp_path):
    # the CLI's orbit pose (reference __main__.py:120-142) drives a real render whose
    # 8-bit image is compared with the C oracle's for the same scene and pose, written through the
    # image convention img[H-1-j, i] (camera.py:33-35)
    from oracle import c_oracle
    from rtgs import synthetic
    from rtgs.__main__ import main, orbit_pose
    import scenes
    w, h, fov = 120, 80, 70.0
    theta, phi, r = 0.6, 1.1, 3.2
    out = str(tmp_path / "orbit.ppm")
    assert main(["--synthetic", "3000", "-r", f"{w},{h}", "-f", str(fov), "-d", "16", "--theta", str(theta),
                 "--phi", str(phi), "--radius", str(r), "--cursor", "0.1,-0.05,0.2", "--out", out]) == 0
    data = open(out, "rb").read()
    head = b"P6\n%d %d\n255\n" % (w, h)
    got = np.frombuffer(data[len(head):], np.uint8).reshape(h, w, 3).astype(np.int32)
    pos, quat = orbit_pose(theta, phi, r, (0.1, -0.05, 0.2))
    focal = float(np.float32((h / 2) / np.tan(fov * np.pi / 360)))
    cam = dict(position=scenes.f32_tuple(pos), rotation=scenes.f32_tuple(quat), width=w, height=h, fx=focal, fy=focal)
    arrays = synthetic.make_scene(3000, w / h, 1)
    want = c_oracle.render(arrays, cam, 16)
    img = np.clip(want["sample"].reshape(w, h, 3), 0.0, 1.0)
    want8 = (np.transpose(img, (1, 0, 2))[::-1] * 255.0 + 0.5).astype(np.int32)
    solid = np.transpose((~want["fragile"]).reshape(w, h), (1, 0))[::-1]
    assert np.abs(got - want8)[solid].max() <= 1            # one 8-bit step: rounding of values at x.5
    assert (np.abs(got - want8)[solid] == 0).mean() > 0.99 and want8.max() > 100 and solid.mean() > 0.99


def test_headless_cli_renders_a_ply_and_a_synthetic_scene(tmp_path, capsys):
    # the reference's `uv run start -o scene.ply -r W,H -f fov -s samples -d depth --scale s`
    # (__main__.py:34-142) without the ti.GUI window: same flags, image written instead of shown.
    # Called in-process: a GPU-initialised process must not exec another program here.
    from rtgs.__main__ import main
    ply = os.path.join(U.GOLDEN, "test.ply")
    a, b = str(tmp_path / "peel.ppm"), str(tmp_path / "onepass.ppm")
    common = ["-o", ply, "-r", "96,64", "-f", "60", "-d", "8", "--scale", "8", "--radius", "3"]
    assert main(common + ["-s", "1", "--out", a]) == 0                 # the viewer loop: one peel per call
    assert main(common + ["-s", "1", "--out", b, "--one-pass"]) == 0
    img_a, img_b = open(a, "rb").read(), open(b, "rb").read()
    assert img_a[:2] == b"P6" and len(img_a) == len(img_b) > 96 * 64 * 3
    assert img_a == img_b and any(v != 0 for v in img_a[-96 * 64 * 3:])   # same picture, not black
    c = str(tmp_path / "syn.png")
    assert main(["--synthetic", "2000", "-r", "64,48", "--radius", "3", "--out", c, "--one-pass", "-s", "2"]) == 0
    assert open(c, "rb").read()[:8] == b"\x89PNG\r\n\x1a\n"
    assert "Gaussians" in capsys.readouterr().out
