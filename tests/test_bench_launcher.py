"""bench.py's self-launch for --gpus N > 1: the argument / environment / exit-status
plumbing of spawn_ranks, driven with stand-in rank scripts -- no GPU, no torch in the children.
The real thing (N ranks of the frame loop, one verified line) runs in the -m gpu suite."""

import os
import subprocess
import sys
import textwrap
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _script(tmp_path, body):
    p = tmp_path / "rank.py"
    p.write_text("import os, sys, time\nrank = int(os.environ['RANK'])\n" + textwrap.dedent(body))
    return str(p)


def test_importing_the_launcher_does_not_import_torch():
    # the parent of the ranks must not be able to initialise the GPU: torch is not even loaded
    code = ("import sys; sys.path.insert(0, %r); import bench; "
            "assert 'torch' not in sys.modules and bench.torch is None; print('ok')" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr


def test_every_rank_gets_its_environment_and_the_arguments(tmp_path):
    import bench
    s = _script(tmp_path, """
        keys = ["RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                "HSA_ENABLE_IPC_MODE_LEGACY"]
        open(os.path.join(os.path.dirname(__file__), f"env{rank}.txt"), "w").write(
            " ".join(os.environ[k] for k in keys) + "|" + " ".join(sys.argv[1:]))
        if rank == 0:
            print('{"line": 1}')
    """)
    out = open(tmp_path / "out.txt", "w+")
    rc = bench.spawn_ranks(3, ["--gpus", "3", "--steps", "7"], script=s, stdout=out, timeout=60)
    assert rc == 0
    out.seek(0)
    assert out.read().strip() == '{"line": 1}'              # rank 0's line, relayed once
    ports = set()
    for r in range(3):
        env, argv = open(tmp_path / f"env{r}.txt").read().split("|")
        rank, local, world, lworld, addr, port, ipc = env.split()
        assert (rank, local, world, lworld, addr, ipc) == (str(r), str(r), "3", "3", "127.0.0.1", "0")
        assert argv == "--gpus 3 --steps 7"
        ports.add(port)
    assert len(ports) == 1 and 1024 < int(ports.pop()) < 65536


def test_a_failing_rank_ends_the_others_and_its_status_is_returned(tmp_path):
    import bench
    s = _script(tmp_path, """
        if rank == 1:
            time.sleep(0.3)
            sys.exit(2)            # bench.EXIT_UNVERIFIED, or any failure
        time.sleep(120)            # a peer stuck in a collective
    """)
    t0 = time.monotonic()
    assert bench.spawn_ranks(2, [], script=s, timeout=60) == 2
    assert time.monotonic() - t0 < 30


def test_a_rank_killed_by_a_signal_is_a_failure(tmp_path):
    import bench
    s = _script(tmp_path, """
        import signal
        if rank == 0:
            os.kill(os.getpid(), signal.SIGKILL)
        time.sleep(120)
    """)
    assert bench.spawn_ranks(2, [], script=s, timeout=60) == 128 + 9


def test_ranks_that_hang_are_ended_at_the_timeout(tmp_path):
    import bench
    s = _script(tmp_path, "time.sleep(120)\n")
    t0 = time.monotonic()
    assert bench.spawn_ranks(2, [], script=s, timeout=1.0) == bench.EXIT_TIMEOUT == 124
    assert time.monotonic() - t0 < 30


def test_main_takes_the_launcher_only_when_it_is_not_already_a_rank(tmp_path, monkeypatch):
    import bench
    calls = []
    monkeypatch.setattr(bench, "spawn_ranks", lambda n, argv, **kw: calls.append((n, list(argv), kw)) or 7)
    monkeypatch.setattr(bench, "rank_main", lambda args: calls.append(("rank", args.gpus)) or 0)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert bench.main(["--gpus", "4", "--steps", "3", "--launch-timeout", "99"]) == 7
    assert calls[-1][0] == 4 and calls[-1][1] == ["--gpus", "4", "--steps", "3", "--launch-timeout", "99"]
    assert calls[-1][2]["timeout"] == 99.0
    assert bench.main(["--gpus", "1"]) == 0 and calls[-1] == ("rank", 1)          # N = 1: no launcher
    monkeypatch.setenv("WORLD_SIZE", "4")                                          # torchrun started us
    assert bench.main(["--gpus", "4"]) == 0 and calls[-1] == ("rank", 4)
    with pytest.raises(SystemExit):
        bench.main(["--gpus", "0"])
