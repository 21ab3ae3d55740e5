"""Rank 0 of `world`'s tile-parallel frame loop on ONE GPU: the real loop (F render streams, event
chain, untile) with a gather that makes the real RCCL call for one peer's worth (world-1 process
group), as tools/host_overhead.py.  Prints host and total microseconds per frame for 1, 2, 3 frames
in flight.  usage: python tools/shard_loop.py [cfg ...]   (default c3 c5)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "rt-gaussian-splat-renderer_amd"), ROOT]
import torch, torch.distributed as dist
import rtgs
from rtgs import synthetic
from rtgs.parallel import TileParallelRenderer

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29537")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))


def gather1(tensor, gather_list=None, dst=0, group=None, async_op=False):
    return dist.gather(tensor, gather_list=[gather_list[0]], dst=0, async_op=async_op)


def nogather(tensor, gather_list=None, dst=0, group=None, async_op=False):
    return None


def timeit(fn, n):
    for _ in range(30): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize(); t_all = time.perf_counter() - t0
    return t_host / n * 1e6, t_all / n * 1e6


world = int(os.environ.get("WORLD", "8"))
for name in (sys.argv[1:] or ["c3", "c5"]):
    cfg = synthetic.CONFIGS[name]
    arrays = synthetic.make_config_scene(cfg); cam = synthetic.bench_camera(cfg.width, cfg.height)
    scene = rtgs.Scene.from_arrays(**arrays)
    camera = rtgs.Camera(cam["position"], cam["rotation"], (cfg.width, cfg.height), (cam["fx"], cam["fy"]))
    n = 400 if name != "c5" else 150
    for F, B in ((1, 1), (2, 1), (3, 1), (2, 2), (4, 2), (3, 3), (4, 4)):
        for label, g in (("one-peer RCCL gather", gather1), ("no gather", nogather)):
            r = TileParallelRenderer(scene, camera, depth=cfg.depth, rank=0, world=world, gather_fn=g,
                                     frames_in_flight=F, gather_batch=B)
            host, total = timeit(r.render_frame, n)
            r.finish(); torch.cuda.synchronize()
            print(f"{name} rank 0 of {world}, {F} frame(s) in flight, {B} per gather, {label}: host {host:.1f} us, total {total:.1f} us per frame")
            r.close()
    del scene
dist.destroy_process_group()
