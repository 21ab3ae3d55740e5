"""One NON-destination rank's frame loop in a single process (gather / p2p stubbed): python tools/rank_loop_probe.py c3 63 1
Prints a marker before every phase (so that a GPU fault names its phase) and checks the shard against a direct render."""
import faulthandler, os, sys
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "rt-gaussian-splat-renderer_amd"), ROOT]
import numpy as np, torch, rtgs
from rtgs import synthetic, _native as nat
from rtgs.parallel import HipBackend, Partition, TileParallelRenderer, _no_gather

def say(*a):
    print(*a, flush=True)

cfg = synthetic.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
shares = [float(v) for v in sys.argv[2:]] or [63.0, 1.0]
arrays = synthetic.make_config_scene(cfg); cam = synthetic.bench_camera(cfg.width, cfg.height)
scene = rtgs.Scene.from_arrays(**arrays)
camera = rtgs.Camera(cam["position"], cam["rotation"], (cfg.width, cfg.height), (cam["fx"], cam["fy"]))
world, rank = len(shares), len(shares) - 1
say("constructing rank", rank, "of", world, "shares", shares)
r = TileParallelRenderer(scene, camera, depth=cfg.depth, rank=rank, world=world, partition=shares, gather_fn=_no_gather,
                         p2p_fn=lambda mine, landing: [])
say("deal", r.partition.name, "local tiles", r.local_tiles, "streams", r._comm is not None)
for k in range(6):
    say("frame", k); r.render_frame()
say("finish"); r.finish(); torch.cuda.synchronize()
say("project_only"); r.project_only(); torch.cuda.synchronize()
say("trace_only"); r.trace_only(); torch.cuda.synchronize()
hb = HipBackend(scene, camera, cfg.depth)
want = torch.zeros_like(r.compact[0][0] if r.compact[0].dim() == 5 else r.compact[0])
if r.local_tiles:
    hb.trace(want, 0, 1, compact=True, shard=hb.shard(r.partition.tiles[rank]))
torch.cuda.synchronize()
got = r.compact[0][0] if r.compact[0].dim() == 5 else r.compact[0]
say("shard equal:", bool(torch.equal(got, want)), float(want.abs().sum()))
r.close()
say("done")
