"""A rank of the launcher test (tests/test_bench_launcher.py): what bench.py's ranks do with the environment spawn_ranks gives
them -- rendezvous on MASTER_ADDR:MASTER_PORT, one SUM all-reduce of the raw partial vector through the package's own
`dp.allreduce_partials`, rank 0 prints ONE JSON line -- on the CPU with gloo.  argv[1] = "fail": rank 1 exits with code 3 before the
rendezvous (the launcher must then stop rank 0, which would wait for it forever); "flaky <marker file>": the first launch reports a
failed rendezvous (exit code 75, bench.EX_RENDEZVOUS), which the launcher retries once on a fresh port."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if len(sys.argv) > 1 and sys.argv[1] == "fail":
    if rank == 1:
        sys.exit(3)
    time.sleep(600)
if len(sys.argv) > 2 and sys.argv[1] == "flaky":       # the first launch fails at the rendezvous (exit code 75), the second one works
    if not os.path.exists(sys.argv[2]):
        if rank == 0:
            time.sleep(1.0)
            open(sys.argv[2], "w").close()
            sys.exit(75)
        time.sleep(600)
import torch
import torch.distributed as dist
from easyhybrid_jl_amd import dp

dist.init_process_group("gloo", rank=rank, world_size=world)
buf = torch.tensor([1.0 + rank, 10.0 * (rank + 1), 1.0], dtype=torch.float64)       # [grad | sse | n] of this shard
dp.allreduce_partials(buf)
if rank == 0:
    print(json.dumps({"n_gpus": world, "ranks_seen": dist.get_world_size(), "sum": buf.tolist(), "local_rank": int(os.environ["LOCAL_RANK"]),
                      "launcher": os.environ.get("EH_BENCH_LAUNCHER"), "master": os.environ["MASTER_ADDR"]}), flush=True)
dist.destroy_process_group()
