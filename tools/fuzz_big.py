"""GPU fuzz of the LARGE-scene paths (slab bins, per-tile lists, the two-level scan) under random cameras: N seeded
scenes.big_fuzz_case cases, sharded trace vs the C oracle on sampled pixels.  usage: python tools/fuzz_big.py N [first]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "rt-gaussian-splat-renderer_amd"), ROOT, os.path.join(ROOT, "tests")]
import test_gpu_random as R
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
first = int(sys.argv[2]) if len(sys.argv) > 2 else 12000
worst, bad, frag, pix, t0 = 0.0, [], 0, 0, time.time()
for seed in range(first, first + n):
    if os.environ.get("FUZZ_LOG_SEEDS"):
        print(f"seed {seed}", file=sys.stderr, flush=True)      # (a GPU fault ends the process: the last such line names the case)
    try:
        e, nf, npx = R.run_big_case(seed)
        worst, frag, pix = max(worst, e), frag + nf, pix + npx
        if e > 1e-4:
            bad.append((seed, e))
    except Exception as exc:
        bad.append((seed, repr(exc)[:200]))
    if (seed - first + 1) % 25 == 0:
        print(f"  {seed - first + 1} cases, worst robust-pixel error {worst:.3e}, failures {len(bad)}", flush=True)
print(f"{n} large-scene cases in {time.time() - t0:.1f}s: worst robust-pixel error {worst:.3e}; fragile {frag}/{pix} = {frag / max(pix, 1):.3%}; failures: {bad[:8]}")
