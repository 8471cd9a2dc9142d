"""`python bench.py --gpus N` without a launcher spawns its own N ranks (bench.spawn_ranks) before anything touches a GPU.
CPU: the launcher itself -- environment, rendezvous, rank 0's line, a failing rank -- with gloo ranks standing in for the GPU ones."""
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "helpers", "gloo_rank.py")


def _bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("eh_bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)                      # import only: main() runs under __main__
    return m


@pytest.mark.parametrize("world", [2, 3, 8])
def test_spawn_ranks_runs_a_gloo_world(world):
    rc, out = _bench().spawn_ranks(world, [sys.executable, WORKER], capture=True, timeout=240)
    assert rc == 0, out
    line = json.loads(out.strip().splitlines()[-1])
    assert line["n_gpus"] == world and line["ranks_seen"] == world and line["local_rank"] == 0
    assert line["master"] == "127.0.0.1" and line["launcher"] == "bench.py:spawn_ranks"
    assert line["sum"] == [sum(1.0 + r for r in range(world)), sum(10.0 * (r + 1) for r in range(world)), float(world)]


def test_a_failing_rank_stops_the_others_and_its_code_comes_back():
    t0 = time.monotonic()
    rc = _bench().spawn_ranks(2, [sys.executable, WORKER, "fail"], timeout=120)
    assert rc == 3 and time.monotonic() - t0 < 60


def test_a_failed_rendezvous_is_retried_once_on_a_fresh_port(tmp_path):
    # (advisor, round 3: the free port is found by bind(0) / close and can be taken before rank 0 binds)
    marker = str(tmp_path / "first_try_done")
    rc, out = _bench().spawn_ranks(2, [sys.executable, WORKER, "flaky", marker], capture=True, timeout=240)
    assert rc == 0 and os.path.exists(marker), out
    assert json.loads(out.strip().splitlines()[-1])["ranks_seen"] == 2


def test_importing_bench_does_not_touch_torch_or_the_gpu():
    code = ("import importlib.util, sys; s = importlib.util.spec_from_file_location('b', %r); m = importlib.util.module_from_spec(s); "
            "s.loader.exec_module(m); assert 'torch' not in sys.modules, 'bench.py imported torch at module level'; print('ok')" % os.path.join(ROOT, "bench.py"))
    assert subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120).stdout.strip() == "ok"


def test_gpus_n_without_world_size_takes_the_launcher_branch():
    # the parent of `bench.py --gpus 2` must spawn, not raise "launch with torch.distributed.run": replace the command it would run
    code = ("import importlib.util, sys, os; s = importlib.util.spec_from_file_location('b', %r); m = importlib.util.module_from_spec(s); s.loader.exec_module(m)\n"
            "seen = {}\n"
            "def fake(n, cmd, **kw):\n    seen['n'] = n; seen['cmd'] = cmd; return 0\n"
            "m.spawn_ranks = fake; sys.argv = ['bench.py', '--gpus', '2', '--steps', '7', '--warmup', '1']\n"
            "os.environ.pop('WORLD_SIZE', None)\n"
            "try:\n    m.main()\nexcept SystemExit as e:\n    assert e.code == 0\n"
            "assert seen['n'] == 2 and seen['cmd'][-6:] == ['--gpus', '2', '--steps', '7', '--warmup', '1'] and 'torch' not in sys.modules; print('ok')" % os.path.join(ROOT, "bench.py"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.stdout.strip() == "ok", r.stderr
