"""bench.py as the driver runs it for N > 1: a bare `python bench.py --gpus N` that starts its own
ranks.  On a one-GPU box the ranks share cuda:0 and talk over gloo -- everything of
the frame loop except RCCL itself: F render streams, rtgs_shard_frame / rtgs_frames_land, the
gather, the per-rank statistics, the verification against the single-rank render, the exit status."""

import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(args, env=None, timeout=900):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    e.setdefault("RTGS_TRACE_PHASES", "1")     # a hang or a timeout then names the phase every rank stood in (stderr)
    e.update(env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                         timeout=timeout, env=e, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    return out, (json.loads(lines[-1]) if lines else None)


@pytest.mark.parametrize("extra,env", [([], {}), (["--gather-batch", "4"], {"RTGS_FRAMES_IN_FLIGHT": "3"})])
def test_bare_bench_with_two_ranks_starts_them_itself_and_verifies(extra, env):
    out, line = _run(["--gpus", "2", "--backend", "gloo", "--same-device", "--steps", "6", "--warmup", "2",
                      "--config", "c2", *extra], env)
    assert out.returncode == 0, out.stderr[-8000:]
    assert line is not None and line["n_gpus"] == 2 and line["verified"] is True
    assert line["value"] > 0 and line["steps"] == 6 and line["timed_s"] >= 1.0
    pr = line["per_rank"]
    assert len(pr["trace_ms"]) == 2 and pr["trace_ms_min"] <= pr["trace_ms_max"]
    assert sum(pr["local_tiles"]) == -(-800 // 16) * -(-600 // 16)
    assert "gloo" in line["config"]["parallelism"]            # a rehearsal must say so
    # every deal ran and verified: round 4's two -- the equal round-robin one (dist.gather per frame) and the one
    # calibrated on this machine (unequal shares, explicit tile lists, point to point) -- and round 5's two: B frames
    # per library call and collective, landing on rank 0 and on ROTATING roots (one all_to_all per batch)
    gb = extra[1] if extra else "8"
    parts = line["secondary"]["partitions"]
    assert set(parts) == {"equal/root/B1", "auto/root/B1", f"equal/root/B{gb}", f"equal/rotate/B{gb}", f"wedges/rotate/B{gb}"}
    assert all(p is not None and p["verified"] is True for p in parts.values())
    assert parts["equal/root/B1"]["deal"] == "round-robin" and line["secondary"]["partition_reported"] in parts
    assert parts[f"equal/rotate/B{gb}"]["landing"] == "rotate" and parts[f"equal/rotate/B{gb}"]["frames_per_collective"] == int(gb)
    cal = parts["auto/root/B1"]["calibration"]
    assert cal["t_shard_ms"] > 0 and cal["t_frame_ms"] > cal["t_shard_ms"] * 0.5 and cal["t_exchange_ms"] > 0
    assert abs(sum(parts["auto/root/B1"]["shares"]) - 1) < 1e-6 and parts["auto/root/B1"]["shares"][0] >= 0.5 - 1 / 64
    # a deal that leaves ranks without tiles (the one-GPU rehearsal's calibration keeps everything on rank 0) never
    # carries the headline: the line then says how many ranks rendered
    rep = parts[line["secondary"]["partition_reported"]]
    assert rep["rendering_ranks"] == 2


def test_explicit_unequal_shares_cross_the_point_to_point_exchange():
    # the calibrated deal of a one-GPU gloo rehearsal keeps everything on the destination (the "link" is a host round
    # trip); explicit shares make rank 1 trace a quarter of the tiles and send exactly those bytes
    out, line = _run(["--gpus", "2", "--backend", "gloo", "--same-device", "--steps", "4", "--warmup", "1",
                      "--config", "c2", "--partition", "3,1"])
    assert out.returncode == 0, out.stderr[-2000:]
    assert line["verified"] is True and "weighted 48/16 of 64" in line["config"]["parallelism"]
    tiles = line["per_rank"]["local_tiles"]
    assert sum(tiles) == 1900 and 0.2 < tiles[1] / 1900 < 0.3


def test_the_sector_deal_with_its_region_cull_verifies_through_the_real_exchange():
    out, line = _run(["--gpus", "2", "--backend", "gloo", "--same-device", "--steps", "4", "--warmup", "1",
                      "--config", "c2", "--partition", "wedges"])
    assert out.returncode == 0, out.stderr[-2000:]
    assert line["verified"] is True and "2 wedges" in line["config"]["parallelism"]
    assert line["per_rank"]["local_tiles"] == [950, 950]


@pytest.mark.parametrize("hook", ["RTGS_BENCH_TEST_HANG", "RTGS_BENCH_TEST_RAISE"])
def test_a_deal_that_hangs_or_raises_on_one_rank_does_not_take_the_finished_deals_number_with_it(hook):
    # the unequal deals' point-to-point exchange has never met RCCL: if the second deal never ends (test hook: the last
    # rank sleeps instead of running it) -- or RAISES on the last rank only, which then waits in the deal's collective
    # outcome while rank 0 is still inside the deal's own collectives -- every rank's watchdog fires, rank 0
    # prints the line of the deal that did finish and verify, and EVERY rank ends with status 0 -- in the torchrun form
    # too, where no launcher of ours watches
    out, line = _run(["--gpus", "2", "--backend", "gloo", "--same-device", "--steps", "4", "--warmup", "1",
                      "--config", "c2", "--mode-timeout", "20", "--landing", "root", "--gather-batch", "1"],
                     {hook: "auto/root/B1"}, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert len([l for l in out.stdout.splitlines() if l.startswith("{")]) == 1
    assert line["verified"] is True and line["n_gpus"] == 2
    sec = line["secondary"]
    assert sec["partition_reported"] == "equal/root/B1"
    assert sec["partitions"]["equal/root/B1"]["verified"] is True
    if "partition_abandoned" in sec:          # (the watchdog's line; a raise that every rank saw in step just drops the deal)
        assert sec["partition_abandoned"]["partition"] == "auto/root/B1" and "given up" in out.stderr
        assert "auto/root/B1" not in sec["partitions"]
    else:
        assert hook.endswith("RAISE") and sec["partitions"]["auto/root/B1"] is None


def test_launched_by_torch_distributed_run_it_is_a_rank_as_before():
    # the driver's other launch form: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--backend", "gloo", "--same-device", "--steps", "4", "--warmup", "1",
                          "--config", "c2", "--partition", "equal"], capture_output=True, text=True, timeout=900, env=e, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                     # ONE line, from rank 0
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["verified"] is True
    assert set(line["secondary"]["partitions"]) == {"equal/root/B1", "equal/root/B8", "equal/rotate/B8"}


def test_more_ranks_than_devices_is_refused_with_a_nonzero_status():
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a one-GPU box")
    out, line = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--config", "c2"])
    assert out.returncode != 0 and line is None
    assert "devices are visible" in out.stderr


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
@pytest.mark.parametrize("extra", [[], ["--gather-batch", "2"]])
def test_two_ranks_over_rccl(extra):
    out, line = _run(["--gpus", "2", "--steps", "20", "--warmup", "5", "--config", "c2", *extra])
    assert out.returncode == 0, out.stderr[-2000:]
    assert line["verified"] is True and "RCCL" in line["config"]["parallelism"]


def test_rccl_accepts_the_collectives_as_the_frame_loop_calls_them():
    # No N > 1 hardware has ever met this code; what CAN be checked on one GPU is that RCCL itself -- a one-rank "nccl"
    # group -- takes the exchange calls exactly as rtgs.parallel issues them: all_to_all_single on the batch buffers
    # (equal split of a 5-D tensor; flat tensors with explicit split sizes), gather into unbound views, an empty
    # batch_isend_irecv, the all-reduced verdict.  (Two ranks on one device are refused by RCCL: that is the rehearsal's
    # gloo path above.)
    code = r'''
import os, sys, datetime
sys.path[:0] = [os.path.join(r"%s", "rt-gaussian-splat-renderer_amd"), r"%s"]
import torch, torch.distributed as dist
from rtgs.parallel import _all_to_all
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=120))
send = torch.arange(8 * 5 * 16 * 16 * 3, dtype=torch.float32, device=dev).reshape(8, 5, 16, 16, 3)
recv = torch.zeros_like(send)
_all_to_all(recv, send, None, None, group=None)
torch.cuda.synchronize()
assert torch.equal(recv, send)
flat_in, flat_out = send.reshape(-1), torch.zeros(send.numel() + 7, dtype=torch.float32, device=dev)
_all_to_all(flat_out[:send.numel()], flat_in, [send.numel()], [send.numel()], group=None)
torch.cuda.synchronize()
assert torch.equal(flat_out[:send.numel()], flat_in) and float(flat_out[send.numel():].abs().sum()) == 0.0
gathered = torch.zeros((1, 8, 5, 16, 16, 3), dtype=torch.float32, device=dev)
dist.gather(send, gather_list=list(gathered.unbind(0)), dst=0, async_op=False)
torch.cuda.synchronize()
assert torch.equal(gathered[0], send)
flag = torch.tensor([1], dtype=torch.int32, device=dev)
dist.all_reduce(flag, op=dist.ReduceOp.MIN)
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert int(flag.item()) == 1 and float(t.item()) == 1.5
rows = [None]
dist.all_gather_object(rows, {"rank": 0})
dist.barrier()
dist.destroy_process_group()
print("RCCL_OK")
''' % (ROOT, ROOT)
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=e, cwd=ROOT)
    assert out.returncode == 0 and "RCCL_OK" in out.stdout, out.stderr[-2000:]
