"""-m gpu: the peer-to-peer gradient exchange of the fused step kernel (eh_p2p_*) with two ranks.
A 1-GPU box can only host both ranks on the same device, which still exercises the whole protocol
(IPC-mapped receive buffers, staging + last-workgroup publish, flags, deadline, bitwise-identical
replicas, the fallback negotiation); only the xGMI transport itself is not covered.  The ranks run
in child processes started BEFORE this process touches the GPU (file name sorts ahead of
test_gpu_parity.py): a process that has initialised HIP must not fork+exec on the GPU boxes."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, port, tool="p2p_two_ranks.py"):
    import torch
    if torch.cuda.is_initialized():
        pytest.skip("this process already initialised the GPU; run this file first (or alone)")
    env = dict(os.environ, **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tools", tool)]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in (r.stdout + r.stderr).replace("rank ", "\nrank ").splitlines() if l.startswith("rank ")]
    assert r.returncode == 0, "\n".join(lines) + "\n" + (r.stdout + r.stderr)[-1500:]
    return lines


def test_two_ranks_peer_to_peer_exchange_matches_single_engine_training():
    lines = _run({}, 29561)
    first = [l for l in lines if "max|theta-ref|" in l]
    assert len(first) == 2 and all("p2p=True" in l and "replicas_identical=True" in l for l in first), lines


def test_two_ranks_peer_to_peer_exchange_with_run_time_specialised_kernels():
    # DataParallel(specialize=True): every rank compiles its train / eval / cross-GPU kernels before the first step
    lines = _run({"EH_TOOL_SPECIALIZE": "1"}, 29564)
    first = [l for l in lines if "max|theta-ref|" in l]
    assert len(first) == 2 and all("p2p=True" in l and "jit_kernels=1" in l and "replicas_identical=True" in l for l in first), lines


def test_two_ranks_fall_back_to_the_collective_when_one_rank_fails_the_selftest():
    lines = _run({"EH_TOOL_FAIL_SELFTEST": "1"}, 29562)
    first = [l for l in lines if "max|theta-ref|" in l]
    assert len(first) == 2 and all("p2p=False" in l and "replicas_identical=True" in l for l in first), lines


def test_two_ranks_recover_from_a_missed_exchange():
    # one rank runs a step the other does not: the exchange hits its 2 s deadline, DataParallel.check() moves every rank to the
    # all-reduce exchange and re-broadcasts parameters + optimiser state; replicas must be bitwise identical again afterwards
    lines = _run({"EH_TOOL_FORCE_TIMEOUT": "1"}, 29565)
    rec = [l for l in lines if "forced timeout" in l]
    assert len(rec) == 2 and all("check() -> False" in l and "identical=True" in l and "finite=True" in l for l in rec), lines


def test_two_ranks_distributed_train_front_door():
    # train(model, data, distributed=True): shard + per-shard shuffle + replicated evaluation, with and without input BatchNorm, and a two-target model with per-target losses
    lines = _run({"EH_MAX_BLOCKS": "64"}, 29563, tool="train_two_ranks.py")
    assert len(lines) == 6 and all("results_identical_across_ranks=True" in l for l in lines), lines      # (single target +- BatchNorm, two targets) x two ranks


def test_bench_gpus_2_launches_its_own_ranks_and_prints_a_self_describing_line():
    """`python bench.py --gpus 2` with no launcher: the script spawns its two ranks itself, before anything touches
    a GPU; on this one-GPU box both ranks share device 0 (EH_BENCH_SHARE_GPU=1: gloo carries the collectives, the numbers are
    meaningless), which walks the whole multi-rank flow -- N = 1 reference, exchange negotiation and calibration, timed region, line."""
    import json
    import torch
    if torch.cuda.is_initialized():
        pytest.skip("this process already initialised the GPU; run this file first (or alone)")
    env = dict(os.environ, EH_BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 20 and line["warmup"] == 5 and line["scaling"] == "weak" and line["value"] > 0
    cfg = line["config"]
    assert cfg["ranks_seen"] == 2 and cfg["launcher"] == "bench.py:spawn_ranks" and cfg["parallelism"] == "dp2" and cfg["global_batch"] == 2 * 65536
    assert [d["rank"] for d in cfg["rank_devices"]] == [0, 1]
    assert "gradient_exchange" in cfg and line["n1_reference"]["value"] > 0 and 0 < line["weak_scaling_vs_n1_in_this_run"] < 2
    assert line["roofline"]["bursts_timed"] >= 5
