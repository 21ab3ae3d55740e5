"""The tile-parallel loop of rank 0 of 8 (one-peer RCCL gather, as tools/host_overhead.py) for a kernel
trace:  rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/loop_trace.py ; then
python3 tools/loop_trace.py --analyze OUT  prints one steady-state frame's kernel timeline."""
import os, sys, glob, csv
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[1] == "--analyze":
    f = sorted(glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = list(csv.DictReader(open(f)))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("rtgs::(anonymous namespace)::", "")[:44], r.get("Queue_Id", "?")) for r in rows))
    traces = [i for i, e in enumerate(ev) if "k_trace" in e[2]]
    mid = traces[len(traces) // 2]
    t0 = ev[mid][0]
    per = (ev[traces[-10]][0] - ev[traces[10]][0]) / (len(traces) - 20) / 1e3
    print("k_trace start to k_trace start, mean: %.1f us" % per)
    for e in ev[mid - 1: mid + 12]:
        print("  %8.1f .. %8.1f us  (%5.1f)  q%s  %s" % ((e[0] - t0) / 1e3, (e[1] - t0) / 1e3, (e[1] - e[0]) / 1e3, e[3], e[2]))
    sys.exit(0)
sys.path[:0] = [os.path.join(ROOT, "rt-gaussian-splat-renderer_amd"), ROOT]
import torch, torch.distributed as dist
import rtgs
from rtgs import synthetic
from rtgs.parallel import TileParallelRenderer
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29535")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
cfg = synthetic.CONFIGS["c3"]
arrays = synthetic.make_config_scene(cfg); cam = synthetic.bench_camera(cfg.width, cfg.height)
scene = rtgs.Scene.from_arrays(**arrays)
camera = rtgs.Camera(cam["position"], cam["rotation"], (cfg.width, cfg.height), (cam["fx"], cam["fy"]))
def gather1(tensor, gather_list=None, dst=0, group=None, async_op=False):
    return dist.gather(tensor, gather_list=[gather_list[0]], dst=0, async_op=async_op)
r8 = TileParallelRenderer(scene, camera, depth=16, rank=0, world=8, gather_fn=gather1)
for _ in range(300): r8.render_frame()
r8.finish(); torch.cuda.synchronize()
dist.destroy_process_group()
