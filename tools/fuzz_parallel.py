"""Soak of tests/test_gpu_parallel_fuzz.py's generator: N seeded random tile-parallel render-side cases.  usage: python tools/fuzz_parallel.py N [first]
FUZZ_TWINS=1: run_twin_paths_case (coloured twins: sample_all == shards == both sample() loops); FUZZ_BATCHES=1: run_batch_case (FrameBatchRenderer: every image of every call == its single-frame render) instead."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "rt-gaussian-splat-renderer_amd"), ROOT, os.path.join(ROOT, "tests")]
import test_gpu_parallel_fuzz as F
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
first = int(sys.argv[2]) if len(sys.argv) > 2 else 9000
bad, frames, empty, t0 = [], 0, 0, time.time()
for seed in range(first, first + n):
    if os.environ.get("FUZZ_LOG_SEEDS"):
        print(f"seed {seed}", file=sys.stderr, flush=True)      # (a GPU fault ends the process: the last such line names the case)
    try:
        if os.environ.get("FUZZ_BATCHES"):      # the single-GPU FrameBatchRenderer generator instead
            frames += F.run_batch_case(seed)
            continue
        if os.environ.get("FUZZ_TWINS"):        # coloured twins through the RayTracer's own paths (counts pixels)
            frames += F.run_twin_paths_case(seed)
            continue
        c, local, total = F.run_parallel_case(seed)
        frames += c
        empty += local == 0
    except Exception as exc:
        bad.append((seed, repr(exc)[:200]))
    if (seed - first + 1) % 50 == 0:
        print(f"  {seed - first + 1} cases, {frames} shard frames compared, {empty} empty shares, failures {len(bad)}", flush=True)
print(f"{n} tile-parallel cases in {time.time() - t0:.1f}s: {frames} shard frames bit-equal to their direct renders, {empty} empty shares; failures: {bad[:6]}")
