"""Randomised (hypothesis) forms of the oracle's structural properties: for any
small scene and camera the literal depth-peeling loop equals the one-pass top-K form, the order
of the Gaussians is irrelevant, and a pixel subset equals the full frame.  CPU only."""

import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import rtgs_ref as ref
from rtgs import synthetic


def _case(seed, n, w, h, sh):
    arrays = synthetic.make_scene(n, w / h, seed, sh=sh)
    rng = np.random.default_rng(seed + 1)
    cam = synthetic.bench_camera(w, h)
    # a modest random pose around the box, rounded to fp32 like every test camera
    cam["position"] = tuple(float(np.float32(v)) for v in (rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5), 3.0))
    return arrays, cam


def _args(arrays, cam):
    return (ref.OracleScene(**arrays), cam["position"], cam["rotation"], cam["width"], cam["height"],
            cam["fx"], cam["fy"])


@settings(max_examples=12, deadline=None)
@given(seed=st.integers(0, 10_000), n=st.integers(1, 150), w=st.integers(1, 12), h=st.integers(1, 12),
       depth=st.integers(1, 16), sh=st.booleans())
def test_peel_equals_topk_and_subsets(seed, n, w, h, depth, sh):
    arrays, cam = _case(seed, n, w, h, sh)
    a = ref.render_peel(*_args(arrays, cam), depth=depth)
    b = ref.render_topk(*_args(arrays, cam), depth=depth)
    assert np.abs(a["sample"] - b["sample"]).max() < 1e-12
    assert np.abs(a["attenuation"] - b["attenuation"]).max() < 1e-12
    assert (a["hits"] == b["hits"]).all()
    pix = ref.all_pixels(w, h)[:: max(1, (w * h) // 5)]
    c = ref.render_peel(*_args(arrays, cam), depth=depth, pixels=pix)
    full_index = pix[:, 0] * h + pix[:, 1]
    assert np.array_equal(c["sample"], a["sample"][full_index])


@settings(max_examples=8, deadline=None)
@given(seed=st.integers(0, 10_000), n=st.integers(2, 120))
def test_gaussian_order_is_irrelevant(seed, n):
    arrays, cam = _case(seed, n, 8, 6, False)
    perm = np.random.default_rng(seed).permutation(n)
    shuffled = {k: (None if v is None else v[perm]) for k, v in arrays.items()}
    a = ref.render_peel(*_args(arrays, cam), depth=16)
    b = ref.render_peel(*_args(shuffled, cam), depth=16)
    assert np.abs(a["sample"] - b["sample"]).max() < 1e-12
