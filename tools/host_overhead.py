"""Measure host-side cost per frame of the tile-parallel loop on one GPU (emulating rank 0 of 8)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "rt-gaussian-splat-renderer_amd"), ROOT]
import torch, torch.distributed as dist
import rtgs
from rtgs import synthetic
from rtgs.parallel import TileParallelRenderer

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
cfg = synthetic.CONFIGS["c3"]
arrays = synthetic.make_config_scene(cfg); cam = synthetic.bench_camera(cfg.width, cfg.height)
scene = rtgs.Scene.from_arrays(**arrays)
camera = rtgs.Camera(cam["position"], cam["rotation"], (cfg.width, cfg.height), (cam["fx"], cam["fy"]))

def timeit(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize(); t_all = time.perf_counter() - t0
    return t_host / n * 1e6, t_all / n * 1e6

r = TileParallelRenderer(scene, camera, depth=16, rank=0, world=1)
b = r.backend
T = b.tile
per = (r.total_tiles + 7) // 8
compact = b.alloc(per, T, T, 3)
gathered = b.alloc(1, per, T, T, 3)
print("trace 1/8 shard (project+trace): host %.1f us, total %.1f us" % timeit(lambda: b.trace(compact, 0, 8, compact=True)))
print("trace 1/8 shard (trace only):    host %.1f us, total %.1f us" % timeit(lambda: b.trace(compact, 0, 8, compact=True, projected=True)))
print("project only:                    host %.1f us, total %.1f us" % timeit(lambda: b.project()))
img = b.alloc(cfg.width, cfg.height, 3)
g8 = b.alloc(8, per, T, T, 3)
print("untile:                          host %.1f us, total %.1f us" % timeit(lambda: b.untile(g8, 8, per, img)))
def gat():
    w = dist.gather(compact, gather_list=[gathered[0]], dst=0, async_op=True); w.wait()
print("dist.gather world=1 (+wait):     host %.1f us, total %.1f us" % timeit(gat))
def agt():
    w = dist.all_gather_into_tensor(gathered, compact, async_op=True); w.wait()
print("all_gather_into_tensor world=1:  host %.1f us, total %.1f us" % timeit(agt))
def agt_sync():
    dist.all_gather_into_tensor(gathered, compact)
print("  same, async_op=False:          host %.1f us, total %.1f us" % timeit(agt_sync))
def gat_sync():
    dist.gather(compact, gather_list=[gathered[0]], dst=0)
print("dist.gather world=1 sync:        host %.1f us, total %.1f us" % timeit(gat_sync))
print("full frame world=1:              host %.1f us, total %.1f us" % timeit(lambda: r.render_frame()))
# rank 0 of 8, the real frame loop, with a gather that makes the real RCCL call for one peer's worth
# (world-1 process group): host cost and stream structure are the production ones, the 7 transfers are not
def gather1(tensor, gather_list=None, dst=0, group=None, async_op=False):
    return dist.gather(tensor, gather_list=[gather_list[0]], dst=0, async_op=async_op)
for mode in ("1", "0"):
    os.environ["RTGS_RENDER_STREAM"] = mode
    r8 = TileParallelRenderer(scene, camera, depth=16, rank=0, world=8, gather_fn=gather1)
    res = timeit(lambda: r8.render_frame())
    r8.finish()
    print("rank 0 of 8 frame loop, RTGS_RENDER_STREAM=%s: host %.1f us, total %.1f us" % ((mode,) + res))
ev = torch.cuda.Event(); side = torch.cuda.Stream()
print("event.record(stream):            host %.1f us, total %.1f us" % timeit(lambda: ev.record(side)))
print("event.wait():                    host %.1f us, total %.1f us" % timeit(lambda: ev.wait()))
print("stream.wait_event(event):        host %.1f us, total %.1f us" % timeit(lambda: side.wait_event(ev)))
dist.destroy_process_group()
