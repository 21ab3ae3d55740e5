"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol that
include/rtgs_hip.h declares (no compute without a GPU), the pure-host entry points answer, and
the host logic of the rtgs package (synthetic scenes, PLY ingest, vector types) behaves."""

import os
import re

import numpy as np
import pytest
import torch

import scenes
from oracle import rtgs_ref as ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def header_functions():
    text = open(os.path.join(ROOT, "include", "rtgs_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rtgs_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from rtgs import _native as nat
    lib = nat.lib()
    names = header_functions()
    assert len(names) >= 18
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/rtgs_hip.h but not exported"
    assert sorted(nat.PROTOTYPES) == names, "ctypes prototypes and header disagree"


def test_pure_host_entry_points():
    from rtgs import _native as nat
    lib = nat.lib()
    assert lib.rtgs_abi_version() == nat.ABI_VERSION == 7
    assert lib.rtgs_strerror(0) == b"ok"
    for code in (-1, -2, -3, -4, -5, -99):
        assert len(lib.rtgs_strerror(code)) > 0
    assert lib.rtgs_workspace_bytes(0) == 0 and lib.rtgs_workspace_bytes(-5) == 0
    assert lib.rtgs_workspace_bytes(100_000) >= 100_000 * 96
    # the frame-sized figure (ABI v6): 1 KB of per-tile slab list per macro-tile instead of 64 MiB for any frame
    full, hd, uhd = (lib.rtgs_workspace_bytes(100_000), lib.rtgs_workspace_bytes_frame(100_000, 1920, 1080),
                     lib.rtgs_workspace_bytes_frame(100_000, 3840, 2160))
    assert full - hd == (65536 - 120 * 68) * 1024 and uhd - hd == (240 * 135 - 120 * 68) * 1024
    assert lib.rtgs_workspace_bytes_frame(100_000, 0, 5) == 0 and lib.rtgs_workspace_bytes_frame(0, 64, 64) == 0
    assert lib.rtgs_workspace_bytes_frame(100_000, 16 * 4096, 16 * 17) < hd       # too many tiles: no per-tile lists
    t = lib.rtgs_tile_size()
    assert t in (8, 16, 32) and t == nat.tile_size()
    assert lib.rtgs_num_tiles(1920, 1080) == -(-1920 // t) * -(-1080 // t)
    assert lib.rtgs_num_tiles(1, 1) == 1 and lib.rtgs_num_tiles(0, 5) == 0
    total = lib.rtgs_num_tiles(1920, 1080)
    assert sum(lib.rtgs_num_local_tiles(1920, 1080, r, 8) for r in range(8)) == total
    assert lib.rtgs_num_local_tiles(1920, 1080, 0, 1) == total
    assert lib.rtgs_num_local_tiles(t, t, 3, 8) == 0        # more ranks than tiles
    assert lib.rtgs_num_local_tiles(t, t, 0, 0) == 0
    # rtgs_launch_order: host-only; bad arguments are refused, a one-tile image has one order
    import numpy as np
    one = np.full(4, -7, dtype=np.int32)
    assert lib.rtgs_launch_order(t, t, 0, 1, 1, one.ctypes.data) == 1 and one.tolist() == [0, -7, -7, -7]
    assert lib.rtgs_launch_order(1920, 1080, 0, 1, 1, None) == -1
    assert lib.rtgs_launch_order(1920, 1080, 0, 1, 2, one.ctypes.data) == -1      # unknown kind
    assert lib.rtgs_launch_order(1920, 1080, 0, 0, 1, one.ctypes.data) == -1      # no tiles
    assert lib.rtgs_accel_bytes(0) == 0 and lib.rtgs_accel_bytes(100_000) >= 100_000 * 52


def test_argument_validation_needs_no_gpu():
    # every entry point rejects NULL / non-positive sizes before touching the device
    from rtgs import _native as nat
    lib = nat.lib()
    assert lib.rtgs_covariance(0, None, None, None, None, None) == -1
    assert lib.rtgs_generate_rays(None, None, None) == -1
    assert lib.rtgs_fill(None, 4, 0.0, None) == -1
    assert lib.rtgs_quat_op(7, 1, None, None, None, None) == -1
    assert lib.rtgs_untile(None, 1, 1, 4, 4, None, None) == -1
    assert lib.rtgs_event_create(None) == -1 and lib.rtgs_event_destroy(None) == -1 and lib.rtgs_event_query(None) == -1
    assert lib.rtgs_shard_frame(None, None, None, None, None, None, None, 0, None, None, None, None, 0) == -1
    assert lib.rtgs_frames_land(None, 0, 1, 1, 0, 4, 4, None, None, None) == -1
    assert lib.rtgs_frames_land(None, 1, 1, 2, 3, 4, 4, None, None, None) == -1       # n_land > batch
    cam = nat.make_camera((0, 0, 0), (0, 0, 0, 1), 64, 64, 64.0, 64.0)
    assert lib.rtgs_trace(None, cam, None, None, None, None, 1, None, 0, 0, None) == -1
    # peel lists (ABI v7): a count + 16 slots x (index, exponent, t1) per thread of every macro-tile
    t = lib.rtgs_tile_size()
    assert lib.rtgs_peel_list_bytes(1920, 1080) == lib.rtgs_num_tiles(1920, 1080) * t * t * (4 + 16 * 12)
    assert lib.rtgs_shard_frames(None, None, 1, None, None, None, None, None, 0, None, 1, 0, None, None, None, 0) == -1
    assert lib.rtgs_peel_list_bytes(0, 8) == 0
    assert lib.rtgs_trace_lists(None, cam, None, None, 0, None, 0, 0, None) == -1
    assert lib.rtgs_peel(None, cam, None, 0, 0, 0, None, None, None) == -1
    assert lib.rtgs_peel_start(cam, None, 0, 0, None, 1, None) == -1
    with pytest.raises(ValueError):
        nat.check(-2, "depth")
    with pytest.raises(nat.RtgsError):
        nat.check(-4, "launch")


def test_device_tensors_are_required():
    import torch
    from rtgs import _native as nat
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        nat.require_device(torch.zeros(4), "x")


def test_synthetic_scene_is_deterministic_and_as_specified():
    from rtgs import synthetic
    a = synthetic.make_scene(500, 16 / 9, 1003)
    b = synthetic.make_scene(500, 16 / 9, 1003)
    assert all(np.array_equal(a[k], b[k]) for k in a if a[k] is not None)
    assert all(v.dtype == np.float32 for v in a.values() if v is not None)
    assert np.abs(a["position"][:, 0]).max() <= 16 / 9 and np.abs(a["position"][:, 1:]).max() <= 1
    assert np.linalg.norm(a["rotation"].astype(np.float64), axis=1) == pytest.approx(np.ones(500), abs=1e-6)
    s0 = np.sqrt(8.727 * (16 / 9) / 500)
    assert a["scale"].min() >= 0.5 * s0 * (1 - 1e-6) and a["scale"].max() <= 2 * s0 * (1 + 1e-6)
    assert 0.1 <= a["opacity"].min() and a["opacity"].max() <= 0.9
    assert a["sh"] is None and synthetic.make_scene(10, 1.0, 1, sh=True)["sh"].shape == (10, 15, 3)
    # the algorithmic byte count of SURVEY.md 8(d)
    assert synthetic.bytes_per_ray(100_000) == 56_518 and synthetic.bytes_per_ray(10_000) == 5_893
    assert synthetic.bytes_per_ray(1_000_000, sh=True) == 565_648
    cfg = synthetic.CONFIGS["c3"]
    assert (cfg.n, cfg.width, cfg.height, cfg.rays) == (100_000, 1920, 1080, 2_073_600)


def test_ply_reader_matches_the_oracle_reader():
    from rtgs.scene import read_ply
    mine = read_ply(os.path.join(GOLDEN, "test.ply"))
    theirs = ref.read_ply_vertices(os.path.join(GOLDEN, "test.ply"))
    assert list(mine) == list(theirs)
    assert all(np.array_equal(mine[k], theirs[k]) for k in mine)
    with pytest.raises(ValueError):
        read_ply(os.path.join(GOLDEN, "c1_full.npz"))


def test_vector_types():
    from rtgs.utils.types import vec2i, vec3, vec4
    v = vec3(1, 2, 3)
    assert (v.x, v.y, v.z) == (1, 2, 3) and v == vec3([1, 2, 3]) and v != vec3(0)
    assert vec3(0) == (0, 0, 0) and vec4(0, 0, 0, 1).w == 1
    assert vec2i((960, 540)).x == 960 and vec2i((960, 540)).dtype == np.int32
    with pytest.raises(ValueError):
        vec3(1, 2)


def test_records_have_the_reference_defaults():
    # reference tests/test_ray.py:15-28 and tests/test_gaussian.py:16-39 (host constructors)
    import math
    from rtgs.gaussian import Gaussian, new_gaussian
    from rtgs.ray import Ray, new_ray
    from rtgs.utils.types import vec3, vec4
    r = new_ray()
    assert r.origin == vec3(0, 0, 0) and r.direction == vec3(0, 1, 0) and r.start == 0 and r.end == math.inf
    r = new_ray(vec3(1, 2, 3), vec3(0, 0, 1), 1, 2)
    assert r.origin == vec3(1, 2, 3) and r.direction == vec3(0, 0, 1) and (r.start, r.end) == (1, 2)
    z = Ray()
    assert z.origin == vec3(0) and z.direction == vec3(0) and z.start == 0 and z.end == 0
    g = new_gaussian()
    assert g.position == vec3(0) and g.rotation == vec4(0, 0, 0, 1) and g.scale == vec3(1, 1, 1)
    assert g.color == vec3(1, 0, 1) and g.opacity == 1 and g.sh_36 == vec3(0)
    g = new_gaussian(vec3(1, 2, 3), vec4(0, 0, 0.5, 0.5), vec3(2, 3, 4), vec3(1, 0, 0), 0.75)
    assert g.scale == vec3(2, 3, 4) and g.opacity == 0.75
    z = Gaussian()
    assert z.rotation == vec4(0) and z.scale == vec3(0) and z.opacity == 0
    if not torch.cuda.is_available():          # no CPU implementation behind any Gaussian method
        with pytest.raises(RuntimeError, match="HIP device"):
            g.bounding_box()


def test_cli_orbit_pose_and_image_writer(tmp_path):
    # reference __main__.py:120-142: theta=0, phi=pi/2, r=1 puts the camera at (1,0,0) looking at
    # the origin with world Z up; camera axes: +x right, +y up, -z forward (coordinate-system.md)
    import zlib
    from rtgs.__main__ import orbit_pose, write_image
    pos, q = orbit_pose(0.0, np.pi / 2, 1.0)
    assert pos == pytest.approx([1, 0, 0], abs=1e-12) and np.linalg.norm(q) == pytest.approx(1)
    fwd = ref.quat_rot_vec3(q, (0, 0, -1))
    up = ref.quat_rot_vec3(q, (0, 1, 0))
    right = ref.quat_rot_vec3(q, (1, 0, 0))
    assert fwd == pytest.approx([-1, 0, 0], abs=1e-12) and up == pytest.approx([0, 0, 1], abs=1e-12)
    assert right == pytest.approx([0, 1, 0], abs=1e-12)
    for th, ph in [(0.7, 1.1), (2.5, 0.4), (-1.0, 2.6)]:
        p, q = orbit_pose(th, ph, 2.0, (0.1, 0.2, 0.3))
        f = ref.quat_rot_vec3(q, (0, 0, -1))
        assert f == pytest.approx(-(p - np.array([0.1, 0.2, 0.3])) / 2.0, abs=1e-9)
    disp = np.zeros((3, 2, 3), np.float32)       # (W, H, 3), (0,0) bottom-left
    disp[0, 0] = (1, 0, 0)
    write_image(str(tmp_path / "a.ppm"), disp)
    data = (tmp_path / "a.ppm").read_bytes()
    assert data.startswith(b"P6\n3 2\n255\n")
    px = np.frombuffer(data[len(b"P6\n3 2\n255\n"):], np.uint8).reshape(2, 3, 3)
    assert px[1, 0].tolist() == [255, 0, 0] and px[0].sum() == 0    # bottom-left pixel is in the LAST row
    write_image(str(tmp_path / "a.png"), disp)
    png = (tmp_path / "a.png").read_bytes()
    assert png.startswith(b"\x89PNG\r\n\x1a\n") and b"IEND" in png


def test_driver_entry_points_exist_and_fail_loudly_without_a_gpu():
    # the round driver imports __graft_entry__ (build / smoke) and runs bench.py; without a HIP
    # device both must say so instead of falling back to anything
    import importlib.util
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("graft_entry", os.path.join(root, "__graft_entry__.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert callable(mod.build) and callable(mod.smoke)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--help"], capture_output=True, text=True)
    assert out.returncode == 0 and "--gpus" in out.stdout and "--steps" in out.stdout and "--warmup" in out.stdout
    import torch
    if not torch.cuda.is_available():
        run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0"],
                             capture_output=True, text=True)
        assert run.returncode != 0 and "HIP device" in (run.stderr + run.stdout)


def test_bench_uses_profiled_counters_only_for_the_build_they_were_profiled_on(tmp_path, monkeypatch):
    # roofline.frac divides a counter from profiles/ by a live kernel time; the summary
    # carries the hash of the sources it was profiled on (build.source_hash, written by
    # tools/summarize_prof.py) and bench.py refuses counters of any other build
    import importlib.util
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    h, count_lib = bench.source_hash()
    assert len(h) == 16 and int(h, 16) >= 0 and count_lib.endswith("librtgs_hip_count.so")
    assert bench.source_hash()[0] == h                      # a function of the files alone
    prof = tmp_path / "profiles"
    prof.mkdir()
    summary = {"source_hash": h, "kernels": {"k_trace": {"pmc_mean_per_launch": {"SQ_INSTS_VALU": 8.0e7, "SQ_INSTS_LDS": 1.0e6}}},
               "k_trace_hbm_bytes_per_launch": {"traffic_upper": 5.0e7}}
    (prof / f"{bench.PROFILE_ROUND}_c3_summary.json").write_text(json.dumps(summary))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    got = bench.profiled_counters("c3", 1, h)
    assert got["valu"] == 8.0e7 and got["lds"] == 1.0e6 and got["traffic"] == 5.0e7 and got["stale"] is None
    stale = bench.profiled_counters("c3", 1, "0" * 16)
    assert stale["valu"] is None and stale["traffic"] is None and "source hash" in stale["stale"]
    assert bench.profiled_counters("c3", 2, h)["valu"] is None          # counters are of a one-GPU launch
    assert all(v is None for v in bench.profiled_counters("c5", 1, h).values())


def test_build_is_keyed_on_the_source_hash_not_on_mtimes(tmp_path):
    # build() used to skip when the .so was newer than the sources; now the stamp
    # beside the library must equal the hash of csrc/* + include/rtgs_hip.h
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("rtgs_build", os.path.join(root, "rt-gaussian-splat-renderer_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    lib = tmp_path / "lib.so"
    lib.write_bytes(b"x")
    assert not b._stamp_ok(str(lib))                         # no stamp: rebuild
    (tmp_path / "lib.so.srchash").write_text("0" * 16 + "\n")
    assert not b._stamp_ok(str(lib))                         # another tree's stamp: rebuild
    (tmp_path / "lib.so.srchash").write_text(b.source_hash() + "\n")
    assert b._stamp_ok(str(lib))
    # the flags are part of the stamp -- a diagnostic build must not pass for the product library
    assert not b._stamp_ok(str(lib), ["-DRTGS_ABLATE=1"])
    (tmp_path / "lib.so.srchash").write_text(b._stamp(["-DRTGS_ABLATE=1"]) + "\n")
    assert b._stamp_ok(str(lib), ["-DRTGS_ABLATE=1"]) and not b._stamp_ok(str(lib))
    # what build() left in the tree is current: the product library without flags, the counting build with its own
    assert b._stamp_ok(b.LIB) and b._stamp_ok(b.COUNT_LIB, ["-DRTGS_COUNT=1"]) and not b._stamp_ok(b.COUNT_LIB)


# every compile-time switch csrc/rtgs_trace.hip still has (round 5: twelve -- STAMPS, COUNT, ABLATE, BINS, TILE_BINS, HORIZON,
# LOOKAHEAD, TIER, SH_STAGE and the three launch bounds WAVES_PER_SIMD / _MAIN / _SH), each in a non-default setting
NON_DEFAULT_BUILDS = [["-DRTGS_STAMPS=1"], ["-DRTGS_COUNT=1"], ["-DRTGS_BINS=0"], ["-DRTGS_TILE_BINS=0"],
                      ["-DRTGS_ABLATE=1"], ["-DRTGS_ABLATE=2"], ["-DRTGS_ABLATE=3"], ["-DRTGS_ABLATE=4"],
                      ["-DRTGS_HORIZON=0"], ["-DRTGS_LOOKAHEAD=0"], ["-DRTGS_LOOKAHEAD=2"],
                      ["-DRTGS_TIER=4"], ["-DRTGS_SH_STAGE=0"], ["-DRTGS_WAVES_PER_SIMD_MAIN=6", "-DRTGS_WAVES_PER_SIMD_SH=4", "-DRTGS_WAVES_PER_SIMD=3"],
                      ["-DRTGS_BINS=0", "-DRTGS_TILE_BINS=0", "-DRTGS_STAMPS=1", "-DRTGS_COUNT=1"]]


def test_every_remaining_kernel_switch_still_compiles():
    # build() only compiles the default combination (+ the counting build); the diagnostic and
    # fallback forms the tools use must not rot.  Front end + template instantiation of every instantiation for
    # gfx950 (-fsyntax-only, ~2 s each), in parallel; the forms three rounds ruled out are gone from the source.
    import shutil
    import subprocess
    from concurrent.futures import ThreadPoolExecutor
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = os.path.join(ROOT, "rt-gaussian-splat-renderer_amd", "csrc")

    def check(flags):
        out = subprocess.run([hipcc, "-O3", "-fno-slp-vectorize", "--offload-arch=gfx950", "-std=c++17", "-fsyntax-only", "-Wno-unused-command-line-argument",
                              "-I" + os.path.join(ROOT, "include"), "-I" + src, *flags, os.path.join(src, "rtgs_trace.hip")],
                             capture_output=True, text=True, timeout=300)
        return flags, out.returncode, out.stderr[-1500:]

    with ThreadPoolExecutor(4) as pool:
        for flags, rc, err in pool.map(check, NON_DEFAULT_BUILDS):
            assert rc == 0, f"{' '.join(flags)} does not compile:\n{err}"
    text = open(os.path.join(src, "rtgs_trace.hip")).read()
    for gone in ("RTGS_MED3", "RTGS_TWO_LEVEL", "RTGS_SH_ABLATE", "RTGS_ABLATE_PAYLOAD", "RTGS_SKIP_LOWER", "RTGS_FAST_DIV",
                 "RTGS_COMPOSITE_EXIT", "k_super_bbox(", "RTGS_SH_FMA", "RTGS_STAGE2",
                 # round 5: the two-waves-per-tile form, the traversal cut, the one-entry list, the projection on a stream
                 # of its own, and the tuning macros whose sweeps stood
                 "kSplit", "kCut", "RTGS_CUT_TRAVERSAL", "RTGS_SPLIT_MAX_TILES", "rtgs_frame_ahead(", "list_insert_par(key",
                 "RTGS_SLAB_ITERS", "RTGS_GROUP", "RTGS_SLAB_CAP", "RTGS_SLABS_PER_WAVE", "RTGS_EXPAND", "RTGS_BIG_SLABS",
                 "RTGS_SUPER", "RTGS_SORT_MAX", "RTGS_BIN_GROUPS", "RTGS_BIN_ITERS", "RTGS_BINS_MIN_SLABS"):
        assert gone not in text, gone
    # what is left: every RTGS_ macro the kernels test is one of the twelve switches (or an ABI constant of rtgs_hip.h)
    import re
    switches = set(re.findall(r"#ifndef (RTGS_[A-Z_0-9]+)", text))
    assert switches == {"RTGS_STAMPS", "RTGS_COUNT", "RTGS_ABLATE", "RTGS_WAVES_PER_SIMD", "RTGS_WAVES_PER_SIMD_MAIN",
                        "RTGS_WAVES_PER_SIMD_SH", "RTGS_LOOKAHEAD", "RTGS_BINS", "RTGS_HORIZON", "RTGS_TIER", "RTGS_SH_STAGE",
                        "RTGS_TILE_BINS"}, switches
    # and twelve instantiations of k_trace: {SH, no SH} x {<= 4096 slabs, more} x {whole frame, peel lists, batched frames}
    assert len(re.findall(r"k_trace<HAS_SH, BIG, (?:true|false), (?:true|false)>", text)) == 3
