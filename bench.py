#!/usr/bin/env python3
"""Benchmark of the ray-traced Gaussian-splat render path (BASELINE.json metric: Mrays/s at
1920x1080, 100k Gaussians; achieved HBM GB/s vs peak).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3]

A step = one full frame of the hot path: per-frame projection + trace/sort/composite of every
pixel (depth 16) and, for N > 1, the RCCL gather of the tile shards and the de-interleave on
rank 0.  Scene and framebuffers are resident in HBM when the timed region starts.  One JSON
line is printed by rank 0.

N > 1: one rank per GPU.  Launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* in the environment) the process IS a rank; launched bare (`python bench.py --gpus N`)
it starts its N ranks itself -- child processes created BEFORE this process has touched the GPU
(torch is not even imported in the parent), whose rank 0 prints the line -- and exits with their
status: non-zero when a rank failed, timed out or the frames did not verify.
"""

from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "rt-gaussian-splat-renderer_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

# numpy / torch / torch.distributed are imported by rank_main(): the launcher half of this file
# (spawn_ranks) must stay free of anything that could initialise the GPU in the parent
np = torch = dist = None

HBM_PEAK_GBS = 8000.0  # MI355X spec peak, /opt/skills/guides/MI355X_MICROARCH.md
# VALU issue peak: 256 CUs x 4 SIMDs, one wave64 VALU instruction per SIMD every 2 cycles
# (SIMD-32: MI355X_MICROARCH.md "v_fma_f32 (wave64) 2 cyc"), at the 2.4 GHz peak engine clock.
# (Inside k_trace the clock measures 2.14 GHz -- s_memtime against the 100 MHz s_memrealtime,
# tools/stamps.py, profiles/r02_phase_stamps.txt; the fraction below is against the 2.4 GHz peak.)
N_SIMD = 1024
ENGINE_CLOCK_HZ = 2.4e9
VALU_PEAK_GINST = N_SIMD * ENGINE_CLOCK_HZ / 2 / 1e9   # G wave-instructions / s
PROFILE_ROUND = "r05"
MIN_TIMED_S = 1.0           # a timed batch shorter than this is repeated and the MEDIAN batch is reported
MAX_BATCHES = 1600
LDS_BYTES_PER_INST = 256    # 64 lanes x 4 B: k_trace's LDS instructions are b32 (upper bound for its few b128 reads: x4)
N_CU = 256
# vector L1 (TCP): what the per-lane SH gathers of the gathering SH forms run into.  Measured ceiling
# (tools/ubench/l1_tag_rate.hip, profiles/r04_ubench_l1_tag_rate.txt): TCP_TOTAL_CACHE_ACCESSES_sum advances 0.97-1.00 per
# clock per CU for uniform 16-B-per-lane gathers of an L1-resident table (16 per instruction at least: one per 4 lanes) and
# 1.25 for the SH-like pattern (~6 records per 16 lanes, same offset) with the TA 99 % busy; an access moves up to 64 B.
L1_PEAK_ACCESSES_PER_CLK = 1.25
L1_PEAK_GBS = N_CU * 64 * L1_PEAK_ACCESSES_PER_CLK * ENGINE_CLOCK_HZ / 1e9
LDS_PEAK_GBS = N_CU * 128 * ENGINE_CLOCK_HZ / 1e9


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c3", help="c2 | c3 | c5 (SURVEY.md 8d)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-frame-batch", action="store_true", help="N = 1: only the one-frame-per-launch loop (profiling runs: "
                    "the batched launches would be averaged into k_trace's counters)")
    ap.add_argument("--min-timed-s", type=float, default=MIN_TIMED_S, help="the K-step batch is repeated until this much has "
                    "been timed (profiling runs under rocprofv3 --pmc, where every launch is serialised, pass less)")
    ap.add_argument("--no-scene-hit", action="store_true", help="skip the arbitrary-ray Scene.hit figure")
    ap.add_argument("--no-secondary", action="store_true", help="headline only: no Scene.hit and no t_cutoff "
                    "figures (profiling runs: their kernels would be averaged into k_trace's counters)")
    ap.add_argument("--cpu-rays", type=int, default=256, help="rays of the NumPy baseline sample")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo "
                    "only for functional rehearsal of the N > 1 path)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: all ranks on cuda:0")
    ap.add_argument("--partition", default="both", help="N > 1: the tile deal -- equal (round-robin, one gather), auto "
                    "(shares calibrated on this machine, point to point), both = time both and report the faster "
                    "verified one; wedges = one compact image sector per rank, its projection culled to it (all = "
                    "the three of them); or explicit per-rank shares, e.g. 0.7,0.3")
    ap.add_argument("--landing", default="both", help="N > 1: where frames land -- root (every frame on rank 0, one gather "
                    "per batch), rotate (frame f on rank f mod N, one all_to_all per batch of frames), both = time both")
    ap.add_argument("--gather-batch", type=int, default=0, help="N > 1: frames per library call and collective for the "
                    "batched deals (0: 8, rounded up to a multiple of N for rotating roots); the root deal is ALSO timed "
                    "frame by frame (batch 1, round 4's loop)")
    ap.add_argument("--dist-timeout", type=float, default=300.0, help="seconds: rendezvous and every collective "
                    "(init_process_group timeout)")
    ap.add_argument("--mode-timeout", type=float, default=0.0, help="N > 1, seconds a tile deal after the first gets before "
                    "the ranks give it up and rank 0 reports the best verified deal so far (0: four times what the "
                    "first deal took, at least 120 s)")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="seconds the self-spawned ranks get "
                    "before the launcher ends them and exits 124")
    return ap.parse_args(argv)


# ---- launcher: `python bench.py --gpus N` with no WORLD_SIZE in the environment -------------------

EXIT_UNVERIFIED = 2      # the N > 1 frames did not equal the single-rank render
EXIT_TIMEOUT = 124


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_env(rank: int, world: int, port: int, base=None) -> dict:
    """The environment torch.distributed.run would give rank `rank` of one node."""
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this driver
    return env


def spawn_ranks(world: int, argv, script=None, python=None, timeout: float = 1500.0, stdout=None, stderr=None,
                poll_s: float = 0.05) -> int:
    """Start `world` fresh child processes `python script *argv`, one per rank, and wait for them.
    Returns 0 when every rank exited 0; otherwise the first failing rank's status (a signal's
    number negated becomes 128 + signal), EXIT_TIMEOUT after `timeout` seconds -- and in both
    cases the ranks still running are ended (their exact PIDs: terminate, then kill).  The caller
    must not have initialised the GPU: children inherit nothing but the environment."""
    script = os.path.abspath(__file__) if script is None else script
    python = sys.executable if python is None else python
    port = _free_port()
    procs = [subprocess.Popen([python, script, *argv], env=rank_env(r, world, port), stdout=stdout, stderr=stderr)
             for r in range(world)]

    def end_all():
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.monotonic() + 10.0
        for p in procs:
            try:
                p.wait(max(0.0, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()

    t0 = time.monotonic()
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                r, c = bad[0]
                print(f"bench.py launcher: rank {r} exited with status {c}; ending the other ranks", file=sys.stderr)
                end_all()
                return c if c > 0 else 128 - c
            if all(c == 0 for c in codes):
                return 0
            if time.monotonic() - t0 > timeout:
                print(f"bench.py launcher: ranks still running after {timeout:.0f} s; ending them", file=sys.stderr)
                end_all()
                return EXIT_TIMEOUT
            time.sleep(poll_s)
    except BaseException:
        end_all()
        raise


CPU_SUBSET = 4096      # "a fixed 4 096-ray random pixel subset (seeded)"


def cpu_baseline(cfg, arrays, cam, n_rays):
    """The oracle timed on this host, on SURVEY 8(d)'s seeded 4 096-pixel subset of the same frame: the C / OpenMP
    restatement (fp64, the closest-hit + midpoint evaluation of the reference, all cores) on all 4 096 pixels, and the
    NumPy restatement (fp32, the reference's literal loop: `depth` closest-hit peels, one core) on the first `n_rays`
    of them -- the NumPy loop does ~16 rays/s at c3, 4 096 rays would be four minutes of the 10-30 s of CPU work the
    bench contract allows.  Reported only."""
    from oracle import rtgs_ref as ref

    rng = np.random.default_rng(4242)
    subset = np.stack([rng.integers(0, cfg.width, CPU_SUBSET), rng.integers(0, cfg.height, CPU_SUBSET)], axis=-1)
    pix = subset[:n_rays]
    osc = ref.OracleScene(**arrays, dtype=np.float32)
    t0 = time.perf_counter()
    ref.render_peel(osc, cam["position"], cam["rotation"], cfg.width, cfg.height, cam["fx"],
                    cam["fy"], depth=cfg.depth, pixels=pix)
    dt = time.perf_counter() - t0
    out = {"value": n_rays / dt / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "port",
           "sample": f"the first {n_rays} of SURVEY 8(d)'s {CPU_SUBSET} seeded random pixels of the {cfg.name} frame, NumPy "
                     f"fp32 literal depth-{cfg.depth} peel loop, {dt:.1f} s, 1 of {os.cpu_count()} cores "
                     "(subset extrapolation); the whole subset is timed through the C restatement: c_openmp"}
    # second line: the C/OpenMP restatement on all cores, the WHOLE 4 096-pixel subset
    try:
        from oracle import c_oracle
        if c_oracle.available():
            t0 = time.perf_counter()
            c_oracle.render(arrays, cam, cfg.depth, subset)
            dt2 = time.perf_counter() - t0
            out["c_openmp"] = {"value": CPU_SUBSET / dt2 / 1e6, "unit": "Mrays/s", "kind": "port",
                               "cores": len(os.sched_getaffinity(0)),
                               "sample": f"all {CPU_SUBSET} pixels of SURVEY 8(d)'s seeded subset, C fp64 one-pass "
                                         f"(oracle/rtgs_oracle.c, OpenMP), {dt2:.1f} s"}
    except Exception as exc:  # the optional line must never break the bench
        out["c_openmp"] = {"error": repr(exc)}
    return out


def closest_hit_rates(scene, cam):
    """Secondary figure: arbitrary-ray Scene.hit throughput on the
    960x540 generated camera rays of the same pose, through the slab-box structure and, once, by
    brute force (same answers: tests/test_gpu_primitives.py)."""
    import rtgs
    w, h = 960, 540
    camera = rtgs.Camera(cam["position"], cam["rotation"], (w, h), (float(h), float(h)))
    camera.generate_ray_field(camera.position, camera.rotation)
    rays = camera.cam_ray_field.data
    scene.accel()
    out = {}
    for name, brute, reps in (("accel", False, 5), ("brute_force", True, 1)):
        scene.hit_batch(rays, brute_force=brute)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            idx, _ = scene.hit_batch(rays, brute_force=brute)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        out[name] = {"rays_per_s": w * h / dt, "ms": dt * 1e3}
    out["rays"] = w * h
    out["hit_fraction"] = float((idx >= 0).float().mean())
    return out


def raytracer_api_times(rtgs, scene, camera, cfg, prefix=""):
    """ms per whole sample through the drop-in class: sample_all(depth); the viewer's pattern of `depth`
    sample(depth) calls (one peel per call) with a STILL camera (the peel lists of the pose are kept: 16 streaming
    peels, no traversal); and the same with a MOVING camera -- a new pose per sample, as the reference's viewer resets
    its sample on every camera event (__main__.py:214-216, 228-230, 248-250: clear_sample(), num_steps = 0), then
    `depth` x sample(depth) (__main__.py:253-256): every sample pays the traversal and the lists again."""
    from rtgs.utils.types import vec3
    tracer = rtgs.RayTracer((cfg.width, cfg.height), scene, camera)
    home = tuple(float(v) for v in camera.position)
    turn = [0]

    def moving_camera_sample():
        turn[0] += 1
        a = 0.05 * turn[0]
        camera.position = vec3((home[0] + 0.2 * float(np.sin(a)), home[1] + 0.1 * float(np.cos(a)), home[2]))
        tracer.clear_sample()
        tracer.num_steps = 0
        for _ in range(cfg.depth):
            tracer.sample(cfg.depth)

    out = {}
    for name, reps, fn in ((prefix + "raytracer_sample_all_ms", 50, lambda: tracer.sample_all(cfg.depth)),
                           (prefix + f"raytracer_sample_x{cfg.depth}_ms", 5,
                            lambda: [tracer.sample(cfg.depth) for _ in range(cfg.depth)]),
                           (prefix + f"raytracer_sample_x{cfg.depth}_new_pose_ms", 10, moving_camera_sample)):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        out[name] = (time.perf_counter() - t0) / reps * 1e3
    camera.position = vec3(home)
    return out


def source_hash():
    """Hash of the sources the library under test was built from (build.py: csrc/* + include/rtgs_hip.h)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("rtgs_build", os.path.join(ROOT, "rt-gaussian-splat-renderer_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.source_hash(), mod.COUNT_LIB


def profiled_counters(cfg_name, world, src_hash):
    """Per-launch k_trace counters from the committed rocprofv3 PMC summary of the same command
    (tools/prof.sh + tools/summarize_prof.py; separate --pmc passes): SQ_INSTS_VALU, SQ_INSTS_LDS
    and the HBM bytes (FETCH_SIZE x2 per the guide's gfx950 correction + WRITE_SIZE).  Counters
    cannot be read from inside the bench, so these are the profiled values -- and they are only
    used when the summary carries the hash of the sources this build was made from: a counter of
    another build divided by this build's time is not a measurement."""
    path = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_{cfg_name}_summary.json")
    out = {"valu": None, "lds": None, "traffic": None, "tcp": None, "ta_busy": None, "lds_busy": None, "lds_conflict": None,
           "source": None, "stale": None}
    if world != 1 or not os.path.exists(path):
        return out
    with open(path) as f:
        summary = json.load(f)
    out["source"] = os.path.relpath(path, ROOT)
    if summary.get("source_hash") != src_hash:
        out["stale"] = f"profile is of source hash {summary.get('source_hash')}, this build is {src_hash}"
        return out
    t = summary.get("k_trace_hbm_bytes_per_launch")
    pmc = summary.get("kernels", {}).get("k_trace", {}).get("pmc_mean_per_launch", {})
    out.update(valu=pmc.get("SQ_INSTS_VALU"), lds=pmc.get("SQ_INSTS_LDS"), traffic=(t["traffic_upper"] if t else None),
               tcp=pmc.get("TCP_TOTAL_CACHE_ACCESSES_sum"))
    if pmc.get("TA_TA_BUSY_sum") and pmc.get("GRBM_GUI_ACTIVE"):
        # busy cycles summed over the 256 TAs / (cycles the GPU was active, summed over the 8 XCDs / 8) / 256
        out["ta_busy"] = pmc["TA_TA_BUSY_sum"] / N_CU / (pmc["GRBM_GUI_ACTIVE"] / 8.0)
    if pmc.get("SQ_LDS_IDX_ACTIVE") and pmc.get("GRBM_GUI_ACTIVE"):
        # rocprofv3's own LdsUtil: cycles the CUs' LDS pipelines were busy / (active cycles x 256 CUs); wide (b128) reads
        # and bank conflicts count for what they occupy, which an instruction count x 256 B does not see
        out["lds_busy"] = pmc["SQ_LDS_IDX_ACTIVE"] / N_CU / (pmc["GRBM_GUI_ACTIVE"] / 8.0)
        out["lds_conflict"] = (pmc.get("SQ_LDS_BANK_CONFLICT") or 0.0) / pmc["SQ_LDS_IDX_ACTIVE"]
    return out


def counted_tests(count_lib, scene, camera, depth, flags=0):
    """One frame through the COUNTING build of the same sources (-DRTGS_COUNT=1, build.py): how many
    ray x Gaussian conic tests k_trace actually executes (candidates a wave evaluated x 64 lanes),
    hits and insertions.  Outside every timed region; the counting build is never timed."""
    import ctypes
    from rtgs import _native as nat
    extra = {"rtgs_debug_counters": (ctypes.c_int32, [ctypes.c_void_p, ctypes.c_int32])}
    cl = nat.load_library(count_lib, extra)
    w, h = camera.buf_size.x, camera.buf_size.y
    out = torch.zeros((w, h, 3), dtype=torch.float32, device="cuda")
    ws = torch.empty(cl.rtgs_workspace_bytes(scene.num_gaussians), dtype=torch.uint8, device="cuda")
    params = nat.RtgsTraceParams(int(depth), 0.0, 0, 1, int(flags), None)
    buf = (ctypes.c_ulonglong * 12)()
    nat.check(cl.rtgs_debug_counters(None, 1), "rtgs_debug_counters")
    nat.check(cl.rtgs_trace(scene.native(), camera.native(), params, nat.ptr(out), None, None, 1, nat.ptr(ws),
                            ws.numel(), 0, nat.stream()), "rtgs_trace (counting build)")
    nat.check(cl.rtgs_debug_counters(ctypes.addressof(buf), 1), "rtgs_debug_counters")
    c = [int(v) for v in buf]
    return {"wave_candidates": c[0], "wave_candidates_with_a_hit": c[1], "insertion_network_runs": c[2],
            "ray_hits": c[3], "ray_insertions": c[4], "gaussian_bounds_tested": c[5], "slab_bounds_tested": c[6],
            "waves": c[7], "tests_executed": c[0] * 64,
            # forms with the SH staging (SH scenes up to 262 k Gaussians): entries shaded from LDS / from memory,
            # distinct Gaussians a workgroup's lists name (mean), workgroups that name more than the 96 staging slots
            "sh_entries_from_lds": c[8], "sh_entries_from_memory": c[9],
            "sh_named_gaussians_per_workgroup": (c[10] / (c[7] / 4)) if c[10] else None, "sh_workgroups_over_slots": c[11]}


def main(argv=None) -> int:
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: be the launcher.  Nothing in this process has touched the
        # GPU (torch is not imported yet); the ranks are fresh children, rank 0 prints the line.
        return spawn_ranks(args.gpus, argv, timeout=args.launch_timeout)
    return rank_main(args)


def rank_main(args) -> int:
    global np, torch, dist
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # N > 1: the caller's stream, two render streams and RCCL's own all want a hardware queue; HIP's default is four
        # per process, beyond which streams share (read when the runtime initialises: set before torch is imported)
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import datetime
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    if args.same_device:
        local_rank = 0
    elif world > 1 and torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} but only {torch.cuda.device_count()} devices are visible "
                         "(--same-device --backend gloo rehearses the N > 1 path on one)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        # a finite timeout: a missing rank or a hung collective ends the run non-zero instead of hanging it
        tmo = datetime.timedelta(seconds=args.dist_timeout)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(args.backend, timeout=tmo)

    import rtgs
    from rtgs import _native as nat
    from rtgs import synthetic
    from rtgs.parallel import TileParallelRenderer
    from rtgs.utils.types import vec3

    cfg = synthetic.CONFIGS[args.config]
    arrays = synthetic.make_config_scene(cfg)
    cam = synthetic.bench_camera(cfg.width, cfg.height)
    scene = rtgs.Scene.from_arrays(**arrays)
    camera = rtgs.Camera(cam["position"], cam["rotation"], (cfg.width, cfg.height), (cam["fx"], cam["fy"]))
    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def verify(renderer):
        """N > 1: consecutive frames with DIFFERENT camera poses are queued back to back through the pipeline (no
        host synchronisation in between); every frame that is held by a rank afterwards -- rank 0's framebuffer for
        the single root, every rank's landed frames of the last batch for rotating roots -- must equal the
        single-rank render of THAT frame's pose bit for bit.  Three sequences: two poses in both orders and an odd
        count (each half of a ring ends a sequence; a stale or swapped buffer shows the other pose), then one that
        ends in a partial batch.  The outcome is all-reduced: every rank returns the same verdict."""
        pose_a, pose_b = (0.15, -0.1, 3.0), tuple(cam["position"])
        rot = tuple(cam["rotation"])
        B = renderer.gather_batch
        sequences = [(pose_a, pose_b), (pose_b, pose_a), (pose_b, pose_a, pose_b),
                     tuple((pose_a, pose_b)[i % 2] for i in range(B + 3)), tuple((pose_b, pose_a)[i % 3 == 0] for i in range(2 * B))]
        held = []                                          # (pose, clone of the landed image)
        for seq in sequences:
            first = renderer._frame
            renderer.render_frames([(p, rot) for p in seq])
            renderer.finish()
            torch.cuda.synchronize()
            if renderer.landing == "rotate":
                last0 = first + (len(seq) - 1) // B * B    # the last batch of the sequence, complete or partial
                mine = [(f, img) for f, img in renderer.landed_frames() if last0 <= f < first + len(seq)]
                want_n = len([f for f in range(last0, first + len(seq)) if renderer.destination(f) == rank])
                if len(mine) != want_n:
                    held.append((None, None))
                held += [(seq[f - first], img.clone()) for f, img in mine]
            elif rank == 0:
                held.append((seq[-1], renderer.framebuffer.clone()))
        ok = True
        if held:
            single = TileParallelRenderer(scene, camera, depth=cfg.depth, rank=0, world=1)
            want = {}
            for p in (pose_a, pose_b):
                camera.position = vec3(p)
                single.render_frame()
                single.finish()
                torch.cuda.synchronize()
                want[p] = single.framebuffer.clone()
            ok = not torch.equal(want[pose_a], want[pose_b])
            for p, got in held:
                ok = ok and p is not None and bool(torch.equal(want[p], got))
        camera.position = vec3(cam["position"])
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())

    class BatchedSingle:
        """N = 1, G whole frames per launch: the adapter run_mode needs around rtgs.parallel.FrameBatchRenderer."""

        def __init__(self, frames, streams):
            from rtgs.parallel import FrameBatchRenderer
            self.fb = FrameBatchRenderer(scene, camera, depth=cfg.depth, frames=frames, streams=streams)
            self.streams = streams
            self.one = TileParallelRenderer(scene, camera, depth=cfg.depth, rank=0, world=1)
            self.gather_batch, self.frames_in_flight, self.landing = self.fb.frames, 1, "batch" if streams == 1 else "batch2"
            self.partition, self.calibration, self.local_tiles = self.one.partition, None, self.one.local_tiles

        def render_frames(self, poses=None, count=None):
            left = int(count)
            while left > 0:
                left -= self.fb.render_frames(count=min(left, self.fb.frames))

        def finish(self):
            pass

        def project_only(self):
            self.one.project_only()

        def trace_only(self):
            self.one.trace_only()

        def verify(self):
            """Eight different poses in one launch: every image must equal its single-frame render bit for bit."""
            p0 = np.asarray(cam["position"], np.float64)
            poses = [(tuple(float(v) for v in p0 + np.array([0.03 * k, -0.02 * k, 0.01 * k])), tuple(cam["rotation"]))
                     for k in range(self.fb.frames)]
            self.fb.render_frames(poses)
            torch.cuda.synchronize()
            ok = True
            for k, (pos, _) in enumerate(poses):
                camera.position = vec3(pos)
                self.one.render_frame()
                torch.cuda.synchronize()
                ok = ok and bool(torch.equal(self.fb.images[k], self.one.framebuffer))
            camera.position = vec3(cam["position"])
            return ok and not torch.equal(self.fb.images[0], self.fb.images[1])

    def run_mode(mode):
        """One deal -- (partition, landing, frames per call and collective) -- end to end: construct (\"auto\"
        calibrates), warm up, the timed batches, the dominant kernel alone, every rank's shard times, the
        verification.  Collective: every rank runs the same modes."""
        from rtgs.parallel import _phase
        partition, landing, gbatch = mode
        _phase("bench: mode", mode)
        if world == 1 and landing in ("batch", "batch2"):
            renderer = BatchedSingle(gbatch, 2 if landing == "batch2" else 1)
        else:
            renderer = TileParallelRenderer(scene, camera, depth=cfg.depth, rank=rank, world=world, partition=partition,
                                            landing=landing, gather_batch=gbatch if world > 1 else None)
        _phase("bench: warm-up")
        renderer.render_frames(count=args.warmup)
        renderer.finish()
        _phase("bench: timed batches")

        def timed_batch():
            """EXACTLY args.steps frames between barrier + synchronize on both sides; MAX over ranks."""
            sync()
            t0 = time.perf_counter()
            renderer.render_frames(count=args.steps)     # (one library call and one collective per batch of the deal)
            renderer.finish()
            sync()
            dt = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([dt], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            return dt

        # a frame is ~0.15 ms: K steps can be a few milliseconds, too thin a sample for one clock reading.
        # The same K-step batch is repeated until MIN_TIMED_S has been timed (the decision uses the
        # all-reduced time: identical on every rank) and the MEDIAN batch is the one reported.
        batches = [timed_batch()]
        while sum(batches) < args.min_timed_s and len(batches) < MAX_BATCHES:
            batches.append(timed_batch())
        elapsed = sorted(batches)[len(batches) // 2]

        # dominant kernel (k_trace) alone, HIP events on the launch stream, projection already done
        _phase("bench: kernel timings")
        reps = max(10, min(args.steps, 100))
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        renderer.project_only()
        renderer.trace_only()
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(reps):
            renderer.trace_only()
        ev1.record()
        torch.cuda.synchronize()
        trace_ms = ev0.elapsed_time(ev1) / reps
        ev0.record()
        for _ in range(reps):
            renderer.project_only()
        ev1.record()
        torch.cuda.synchronize()
        project_ms = ev0.elapsed_time(ev1) / reps
        # every rank's shard trace and projection (ms), so that imbalance between the ranks is visible
        per_rank = None
        if world > 1:
            rows = [None] * world
            dist.all_gather_object(rows, {"rank": rank, "local_tiles": renderer.local_tiles,
                                          "trace_ms": trace_ms, "project_ms": project_ms})
            per_rank = {"trace_ms": [r["trace_ms"] for r in rows], "project_ms": [r["project_ms"] for r in rows],
                        "local_tiles": [r["local_tiles"] for r in rows]}
            for k in ("trace_ms", "project_ms"):
                per_rank[k + "_min"], per_rank[k + "_max"] = min(per_rank[k]), max(per_rank[k])
        _phase("bench: verify")
        verified = verify(renderer) if world > 1 else (renderer.verify() if isinstance(renderer, BatchedSingle) else None)
        _phase("bench: mode done", verified)
        return dict(renderer=renderer, batches=batches, elapsed=elapsed, trace_ms=trace_ms, project_ms=project_ms,
                    per_rank=per_rank, verified=verified, partition=renderer.partition.name, landing=renderer.landing,
                    gather_batch=renderer.gather_batch, rendering_ranks=sum(1 for c in renderer.partition.counts if c),
                    shares=renderer.partition.shares(), calibration=renderer.calibration)

    # N > 1, one run: round 4's loop (equal deal, every frame on rank 0, one gather per frame), the link-aware deal
    # calibrated on this machine (unequal shares, point to point), and round 5's two -- B frames per library call and
    # collective, landing on rank 0 and on ROTATING roots (frame f on rank f mod N, one all_to_all per batch).  The
    # line reports the fastest VERIFIED one, `secondary.partitions` all of them.
    if world == 1:
        # one frame per launch (SURVEY 8(d)'s t_frame: first launch to framebuffer complete; every earlier round's figure)
        # and eight poses per launch of each kernel (rtgs.parallel.FrameBatchRenderer: every frame fully projected, binned
        # and traced into a framebuffer of its own, each verified bit for bit against its single-frame render); `value`
        # is the faster verified one, the other stays under `secondary`
        # (the batches on one stream, or alternating over two so that the next batch's projection runs under this batch's
        # trace: pays for short frames -- c2 -- and not at c3)
        modes = [("equal", "root", None)] + ([] if args.no_frame_batch else [("equal", "batch", 8), ("equal", "batch2", 8)])
    else:
        if "," in args.partition:
            parts = [tuple(float(v) for v in args.partition.split(","))]
            if len(parts[0]) != world:
                raise SystemExit(f"--partition names {len(parts[0])} shares for {world} ranks")
        else:
            parts = {"both": ["equal", "auto"], "all": ["equal", "auto", "wedges"], "equal": ["equal"], "auto": ["auto"],
                     "wedges": ["wedges"]}.get(args.partition)
            if parts is None:
                raise SystemExit("--partition: both | all | equal | auto | wedges | s0,s1,...")
        landings = {"both": ["root", "rotate"], "root": ["root"], "rotate": ["rotate"]}.get(args.landing)
        if landings is None:
            raise SystemExit("--landing: both | root | rotate")
        gb = args.gather_batch if args.gather_batch > 0 else 8
        modes = []
        if "root" in landings:
            modes += [(p_, "root", 1) for p_ in parts]
            modes += [(p_, "root", gb) for p_ in parts if p_ != "auto" and gb > 1]
        if "rotate" in landings:      # (rotating roots: every rank is a destination as often as any other -- no "auto")
            modes += [(p_, "rotate", gb) for p_ in parts if p_ != "auto"]
            if args.partition == "both" and "wedges" not in parts:
                # one compact image sector per rank, its projection culled to it: the fastest deal of the every-rank
                # emulation once frames are batched (profiles/r05_shards.txt)
                modes.append(("wedges", "rotate", gb))
    def emit_line(best, reported, abandoned=None, frame_ms_median=None):
        """Rank 0: the ONE JSON line, from a finished run of a tile deal (`best` = a run_mode result)."""
        renderer, batches, elapsed = best["renderer"], best["batches"], best["elapsed"]
        trace_ms, project_ms, per_rank, verified = best["trace_ms"], best["project_ms"], best["per_rank"], best["verified"]
        ms = elapsed / args.steps * 1e3
        mrays = cfg.rays / (elapsed / args.steps) / 1e6
        bpr = synthetic.bytes_per_ray(cfg.n, cfg.sh, cfg.depth)
        local_rays = renderer.local_tiles * nat.tile_size() ** 2
        streamed = bpr * min(local_rays, cfg.rays) / (trace_ms * 1e-3) / 1e9  # GB/s of the SURVEY 8(d) model
        src_hash, count_lib = source_hash()
        pc = profiled_counters(cfg.name, world, src_hash)
        valu, traffic = pc["valu"], pc["traffic"]
        t_k = trace_ms * 1e-3
        achieved = valu / t_k / 1e9 if valu else None
        hbm_gbs = traffic / t_k / 1e9 if traffic else None
        l1_gbs = pc["tcp"] * 64 / t_k / 1e9 if pc["tcp"] else None
        lds_gbs = pc["lds"] * LDS_BYTES_PER_INST / t_k / 1e9 if pc["lds"] else None
        # the bound reported is the resource with the highest fraction of its peak IN THIS CONFIG: the
        # no-SH instantiations issue VALU instructions, the SH ones wait for the vector L1's tag pipeline
        fractions = {"valu": (achieved / VALU_PEAK_GINST) if achieved else None,
                     "vmem-L1": (l1_gbs / L1_PEAK_GBS) if l1_gbs else None,
                     # (the LDS pipelines' busy share when the profile has it: it sees b128 reads and bank conflicts)
                     "lds": pc["lds_busy"] if pc["lds_busy"] is not None else ((lds_gbs / LDS_PEAK_GBS) if lds_gbs else None),
                     "hbm": (hbm_gbs / HBM_PEAK_GBS) if hbm_gbs else None}
        known = {k: v for k, v in fractions.items() if v is not None}
        bound = max(known, key=known.get) if known else "valu"
        by_bound = {"valu": (achieved, VALU_PEAK_GINST, "G wave64-instr/s"), "vmem-L1": (l1_gbs, L1_PEAK_GBS, "GB/s"),
                    "lds": ((pc["lds_busy"] * LDS_PEAK_GBS) if pc["lds_busy"] is not None else lds_gbs, LDS_PEAK_GBS, "GB/s"),
                    "hbm": (hbm_gbs, HBM_PEAK_GBS, "GB/s")}[bound]
        # SURVEY 8(d) honesty columns (ii) and (iii): tests actually executed vs rays x N, LDS bytes/s
        counts = None
        if world == 1 and os.path.exists(count_lib):
            try:
                counts = counted_tests(count_lib, scene, camera, cfg.depth)
            except Exception as exc:   # the honesty columns must never break the headline
                counts = {"error": repr(exc)}
        tests = counts.get("tests_executed") if counts else None
        # what the issue-slot fraction does not say: of the lanes a wave spends on a candidate, how
        # many had a hit, and of those of an insertion-network run, how many inserted (counting build, same frame)
        lanes = None
        if counts and counts.get("wave_candidates") and counts.get("insertion_network_runs"):
            lanes = {"conic_tests_with_a_hit": counts["ray_hits"] / (64.0 * counts["wave_candidates"]),
                     "insertion_runs_that_insert": counts["ray_insertions"] / (64.0 * counts["insertion_network_runs"])}
        lds_bps = lds_gbs * 1e9 if lds_gbs else None
        base_metric = "Mrays/s at 1920x1080, 100k Gaussians; achieved HBM GB/s vs peak"
        metric = base_metric if (cfg.width, cfg.height, cfg.n) == (1920, 1080, 100_000) else \
            f"Mrays/s at {cfg.width}x{cfg.height}, {cfg.n} Gaussians; achieved HBM GB/s vs peak"
        comm = ""
        if world == 1 and best["landing"] in ("batch", "batch2"):
            comm = (f", {best['gather_batch']} whole frames per launch of each kernel (rtgs_shard_frames: every frame projected, "
                    "binned and traced into a framebuffer of its own" +
                    ("; batches alternate over two streams" if best["landing"] == "batch2" else "") +
                    "; secondary.single_gpu_modes has the one-frame-per-launch loop)")
        if world > 1:
            what = "all_to_all to rotating roots (every rank lands 1/N of the frames)" if best["landing"] == "rotate" else "gather to rank 0"
            comm = f" + RCCL {what}" if args.backend == "nccl" else f" + {args.backend} {what} (rehearsal, not RCCL)"
            comm += (f", {renderer.frames_in_flight} frames in flight per rank, {renderer.gather_batch} frames per library "
                     f"call and collective, tile deal: {best['partition']}")
            if best["rendering_ranks"] < world:
                comm += f" -- ONLY {best['rendering_ranks']} of {world} ranks render"
        line = {
            "metric": metric,
            "value": mrays, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "batches": len(batches), "timed_s": sum(batches),
            "batch_ms_per_step": {"min": min(batches) / args.steps * 1e3, "median": ms,
                                  "max": max(batches) / args.steps * 1e3},
            "config": {"workload": f"{cfg.name}: {cfg.n} random Gaussians (seed {cfg.seed}), "
                                   f"{cfg.width}x{cfg.height}, depth {cfg.depth}, "
                                   f"{'SH deg 3' if cfg.sh else 'no SH'}, camera (0,0,3) f=H",
                       "rays_per_frame": cfg.rays, "gaussians": cfg.n,
                       "parallelism": f"tile-parallel x{world}" + comm},
            # k_trace is a VALU-issue kernel (screen-space culling leaves ~0.03 % of the ray x
            # Gaussian pairs; HBM carries < 10 % of its peak): the bound it can fail is the VALU
            # issue rate.  achieved = SQ_INSTS_VALU per launch (committed profile of THIS source
            # hash and config, else null) / the kernel time measured live here with HIP events on
            # the launch stream.
            "roofline": {"bound": bound, "kernel": "k_trace", "achieved": by_bound[0],
                         "peak": by_bound[1], "unit": by_bound[2],
                         "frac": known.get(bound), "fractions": fractions,
                         "valu_achieved_ginst": achieved, "valu_peak_ginst": VALU_PEAK_GINST,
                         "l1_achieved_gbs": l1_gbs, "l1_peak_gbs": L1_PEAK_GBS, "ta_busy_frac": pc["ta_busy"],
                         "lds_busy_frac": pc["lds_busy"], "lds_bank_conflict_share": pc["lds_conflict"],
                         "tcp_cache_accesses_per_launch": pc["tcp"],
                         "traffic": traffic, "traffic_unit": "HBM bytes per k_trace launch (PMC)",
                         "hbm_achieved_gbs": hbm_gbs, "hbm_peak_gbs": HBM_PEAK_GBS,
                         "hbm_frac": (hbm_gbs / HBM_PEAK_GBS) if hbm_gbs else None,
                         "valu_insts_per_launch": valu, "counters_source": pc["source"],
                         "counters_stale": pc["stale"], "source_hash": src_hash,
                         "kernel_ms": trace_ms, "project_ms": project_ms,
                         "algorithmic_bytes_per_ray": bpr,
                         "speedup_vs_streamed_model": streamed / HBM_PEAK_GBS,
                         "tests_executed": tests, "tests_brute_force": cfg.rays * cfg.n,
                         "cull_rate": (tests / (cfg.rays * cfg.n)) if tests else None,
                         "tests_per_s": (tests / (trace_ms * 1e-3)) if tests else None,
                         "counts": counts, "useful_lane_share": lanes,
                         "lds_bytes_per_s": lds_bps, "lds_insts_per_launch": pc["lds"],
                         "note": "bound = the resource with the highest fraction of its peak among fractions{}: valu = "
                                 "SQ_INSTS_VALU / (1024 SIMDs x 2.4 GHz / 2); vmem-L1 = TCP_TOTAL_CACHE_ACCESSES_sum / "
                                 "(256 CUs x 1.25 accesses/clk x 2.4 GHz), the rate a pure SH-like gather loop reaches "
                                 "(profiles/r04_ubench_l1_tag_rate.txt; x 64 B for the GB/s figures); lds = SQ_LDS_IDX_ACTIVE / (GRBM_GUI_ACTIVE x 256 CUs), rocprofv3's LdsUtil, of the profiled run "
                                 "(without that pass: SQ_INSTS_LDS x 256 B / (256 CUs x 128 B/clk x 2.4 GHz)); "
                                 "hbm = (FETCH_SIZE x 2 + WRITE_SIZE) / 8 TB/s; all null when the "
                                 "committed counters are of another source hash.  speedup_vs_streamed_model = "
                                 "SURVEY 8(d)'s every-tile-streams-every-Gaussian byte count / kernel time / 8 TB/s: "
                                 "not a roofline fraction, the kernel culls that stream instead of performing it.  "
                                 "tests_executed = conic evaluations (wave candidates x 64 lanes) of one frame, "
                                 "counted by the -DRTGS_COUNT=1 build of the same sources; cull_rate = that / "
                                 "(rays x Gaussians).  lds_bytes_per_s = SQ_INSTS_LDS (same profile) x 256 B / "
                                 "kernel time."},
        }
        line["secondary"] = {"closest_hit_queries_per_s": mrays * 1e6 * cfg.depth,
                             "frame_latency_ms_median_hip_events": frame_ms_median,
                             "frames_in_flight": renderer.frames_in_flight}
        if world == 1:
            # both ways of queueing K frames on one GPU; `value` is the faster verified one.  "one frame per launch" is
            # SURVEY 8(d)'s t_frame (first launch to framebuffer complete) and what rounds 1-4 reported
            line["secondary"]["single_gpu_modes"] = {
                str(m): (None if r is None else {"ms_per_step": r["elapsed"] / args.steps * 1e3,
                                            "mrays_per_s": cfg.rays / (r["elapsed"] / args.steps) / 1e6, "verified": r["verified"]})
                for m, r in runs.items()}
            line["secondary"]["mode_reported"] = str(reported)
        if world == 1 and not args.no_secondary:
            # the reference's own API path (ray_tracer.py:39-54, viewer loop __main__.py:253-256):
            # rtgs.RayTracer.sample_all (one call per sample) and sample() x depth (one peel per call)
            line["secondary"].update(raytracer_api_times(rtgs, scene, camera, cfg))
            if cfg.name == "c3":
                # the same three figures with degree-3 SH colours: the kernels every PLY capture takes (scene.py:106-107)
                try:
                    cfg_sh = synthetic.CONFIGS["c3sh"]
                    scene_sh = rtgs.Scene.from_arrays(**synthetic.make_config_scene(cfg_sh))
                    line["secondary"].update(raytracer_api_times(rtgs, scene_sh, camera, cfg_sh, prefix="c3sh_"))
                    del scene_sh
                except Exception as exc:      # a secondary figure must never break the headline
                    line["secondary"]["c3sh_error"] = repr(exc)
            # the transmittance cut-off (extension, off in the headline: the reference has none)
            cut = TileParallelRenderer(scene, camera, depth=cfg.depth, rank=0, world=1, t_cutoff=5e-5)
            for _ in range(3):
                cut.render_frame()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                cut.render_frame()
            torch.cuda.synchronize()
            line["secondary"]["frame_ms_with_t_cutoff_5e-5"] = (time.perf_counter() - t0) / 20 * 1e3
        if world == 1 and not args.no_scene_hit and not args.no_secondary:
            line["secondary"]["scene_hit_960x540"] = closest_hit_rates(scene, cam)
        if verified is not None:
            line["verified"] = verified
        if per_rank is not None:
            line["per_rank"] = per_rank
        if world > 1:
            line["secondary"]["partitions"] = {
                str(m): (None if r is None else {"ms_per_step": r["elapsed"] / args.steps * 1e3,
                                            "mrays_per_s": cfg.rays / (r["elapsed"] / args.steps) / 1e6, "verified": r["verified"],
                                            "deal": r["partition"], "landing": r["landing"], "frames_per_collective": r["gather_batch"],
                                            "rendering_ranks": r["rendering_ranks"], "shares": r["shares"],
                                            "calibration": r["calibration"], "per_rank_trace_ms": r["per_rank"]["trace_ms"]})
                for m, r in runs.items()}
            line["secondary"]["partition_reported"] = str(reported)
            if abandoned is not None:
                line["secondary"]["partition_abandoned"] = abandoned
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, arrays, cam, args.cpu_rays)
        print(json.dumps(line), flush=True)

    def mode_name(m):
        part, landing, gbatch = m
        part = part if isinstance(part, str) else ",".join(f"{v:g}" for v in part)
        if world == 1:
            return ("one frame per launch" if landing not in ("batch", "batch2") else
                    f"{gbatch} frames per launch" + (", batches alternating over 2 streams" if landing == "batch2" else ""))
        return f"{part}/{landing}/B{gbatch}"

    def pick(pool_of):
        """The deal the line reports: the fastest VERIFIED one in which every rank renders (a calibrated deal that
        leaves ranks empty is a single-GPU number under an N-GPU label: ADVICE r4 -- it stays under
        secondary.partitions); failing that the fastest verified, failing that the fastest.  `elapsed`, `verified`
        and the shares are all-reduced / identical on every rank, so every rank picks the same deal by itself."""
        done = {m: r for m, r in pool_of.items() if r is not None}
        for keep in (lambda r: (r["verified"] is not False if world == 1 else r["verified"]) and r["rendering_ranks"] == world,
                     lambda r: r["verified"] is not False if world == 1 else r["verified"], lambda r: True):
            pool = {m: r for m, r in done.items() if keep(r)}
            if pool:
                return min(pool, key=lambda m: pool[m]["elapsed"])
        return None

    runs = {}
    giving_up = threading.Event()      # set by a watchdog that has fired: the main thread then only waits for its os._exit
    emit = [emit_line]     # (what the watchdog below prints with)
    t_first = None
    for mode in modes:
        name = mode_name(mode)
        # A deal must not take an earlier one's number with it -- neither by raising nor by HANGING (the exchanges of
        # the unequal and the rotating deals have never met RCCL): every rank arms the same watchdog; when it fires,
        # rank 0 prints the line of the best verified deal so far and every rank leaves with the same status.  The
        # watchdog stays armed through the deal's collective outcome below: a rank that RAISED is then in
        # that all-reduce while its peers are still inside the deal's collectives, and it is the watchdog -- on every
        # rank, the raiser included -- that ends the mismatch.
        watchdog = None
        if world > 1:
            budget = (args.mode_timeout if args.mode_timeout > 0 else
                      max(120.0, 4.0 * t_first) if t_first is not None else max(600.0, args.dist_timeout))

            def give_up(name=name, budget=budget):
                giving_up.set()
                print(f"bench.py rank {rank}: deal {name!r} made no end in {budget:.0f} s: given up", file=sys.stderr, flush=True)
                good = pick({m: r for m, r in runs.items() if r is not None and r["verified"]})
                if rank != 0:
                    time.sleep(5.0)          # (rank 0 prints first)
                    os._exit(0 if good is not None else EXIT_TIMEOUT)
                if good is None or emit[0] is None:
                    os._exit(EXIT_TIMEOUT)
                emit[0](runs[good], good, abandoned={"partition": name, "after_s": budget})
                os._exit(0)

            watchdog = threading.Timer(budget, give_up)
            watchdog.daemon = True
            watchdog.start()
        t0 = time.perf_counter()
        res, err = None, None
        try:
            try:
                if os.environ.get("RTGS_BENCH_TEST_HANG") == name and rank == world - 1:
                    time.sleep(1e6)                    # (test hook: this deal never finishes on the last rank)
                if os.environ.get("RTGS_BENCH_TEST_RAISE") == name and rank == world - 1:
                    raise RuntimeError("test hook: this deal fails on the last rank")
                res = run_mode(mode)
            except Exception as exc:                   # (a deal that cannot run must not take the others' numbers with it)
                err = exc
                print(f"bench.py rank {rank}: deal {name!r} failed: {exc!r}", file=sys.stderr, flush=True)
            if world > 1:
                # the deal's outcome is COLLECTIVE: if any rank failed, every rank drops it
                flag = torch.tensor([0 if res is None else 1], dtype=torch.int32, device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()) == 0:
                    res = None
        except BaseException:
            # the watchdog is ending the job (a peer that has already left closes this rank's connections, which
            # raises here): the watchdog thread owns the exit status and rank 0's line -- stay out of its way
            while giving_up.is_set():
                time.sleep(1.0)
            raise
        if giving_up.is_set():
            while True:
                time.sleep(1.0)
        if watchdog is not None:
            watchdog.cancel()
        if res is None and len(modes) == 1:
            raise err if err is not None else SystemExit(f"bench.py: deal {name!r} failed on another rank")
        runs[name] = res
        if t_first is None:
            t_first = time.perf_counter() - t0
        if world > 1 and rank == 0 and res is not None and len(modes) > 1:
            # a record on STDERR in case a later deal takes the job down (stdout carries exactly one line, at the end)
            print(f"bench.py: deal {name!r} done: {cfg.rays / (res['elapsed'] / args.steps) / 1e6:.1f} Mrays/s, "
                  f"{res['elapsed'] / args.steps * 1e3:.4f} ms/frame, verified {res['verified']}, tiles {res['partition']}",
                  file=sys.stderr, flush=True)
    choice = [pick(runs)]
    if choice[0] is None:
        raise SystemExit("bench.py: no tile deal ran to completion")
    best = runs[choice[0]]
    renderer, batches, elapsed = best["renderer"], best["batches"], best["elapsed"]
    trace_ms, project_ms, per_rank, verified = best["trace_ms"], best["project_ms"], best["per_rank"], best["verified"]

    # SURVEY 8(d) secondary figures: per-frame HIP-event times (median of 30 frames, first launch
    # to framebuffer complete) and closest-hit queries/s = depth x rays/s
    frame_ms_median = None
    if world == 1:
        # first launch to framebuffer complete, one frame at a time (whole frames are never overlapped on one
        # GPU: two frames in flight on two streams measured 0.1468 -> 0.1733 ms per frame at c3, 1.339 -> 1.369 at c5)
        one = TileParallelRenderer(scene, camera, depth=cfg.depth, rank=0, world=1)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
        for _ in range(3):
            one.render_frame()
        torch.cuda.synchronize()
        for a_, b_ in evs:
            a_.record()
            one.render_frame()
            b_.record()
        one.finish()
        torch.cuda.synchronize()
        frame_ms_median = sorted(a_.elapsed_time(b_) for a_, b_ in evs)[len(evs) // 2]
        one.close()

    if rank == 0:
        emit_line(best, choice[0], frame_ms_median=frame_ms_median)
    status = 0
    if world > 1:
        # every rank leaves with the same status: an unverified run must not look like a measurement
        if verified is not True:          # (all-reduced inside verify(): the same on every rank)
            print(f"bench.py rank {rank}: frames did not verify against the single-rank render", file=sys.stderr)
            status = EXIT_UNVERIFIED
        dist.destroy_process_group()
    return status


if __name__ == "__main__":
    sys.exit(main())
