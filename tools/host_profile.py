"""cProfile of the tile-parallel frame loop's host side (rank 0 of 8, one-peer RCCL gather as in host_overhead.py)."""
import os, sys, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "rt-gaussian-splat-renderer_amd"), ROOT]
import torch, torch.distributed as dist
import rtgs
from rtgs import synthetic
from rtgs.parallel import TileParallelRenderer

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
cfg = synthetic.CONFIGS["c3"]
arrays = synthetic.make_config_scene(cfg); cam = synthetic.bench_camera(cfg.width, cfg.height)
scene = rtgs.Scene.from_arrays(**arrays)
camera = rtgs.Camera(cam["position"], cam["rotation"], (cfg.width, cfg.height), (cam["fx"], cam["fy"]))
def gather1(tensor, gather_list=None, dst=0, group=None, async_op=False):
    return dist.gather(tensor, gather_list=[gather_list[0]], dst=0, async_op=async_op)
r8 = TileParallelRenderer(scene, camera, depth=16, rank=0, world=8, gather_fn=gather1)
for _ in range(50): r8.render_frame()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(2000): r8.render_frame()
pr.disable()
r8.finish(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
dist.destroy_process_group()
