"""Two data-parallel ranks on ONE GPU (the only multi-rank set-up a 1-GPU box allows): exercises the
peer-to-peer exchange of the fused step kernel (eh_p2p_*: IPC-mapped receive buffers, flags, deadline)
end to end and checks the replicas against single-engine training on the union of the two shards.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/p2p_two_ranks.py

gloo carries the handshake (RCCL refuses two ranks on one device); EH_DP_P2P=0 runs the fallback."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
import easyhybrid_jl_amd as eh
from oracle import hybrid_oracle as ho
from tests import util

if os.environ.get("EH_TOOL_HOST_ALLREDUCE", "1") == "1":
    # the fallback exchange of this tool: all-reduce on the host (gloo's own handling of CUDA tensors races with
    # kernels on the legacy default stream in this PyTorch build; RCCL, the production backend, is in-stream)
    def _host_allreduce(buf, group=None):
        if buf.is_cuda:
            torch.cuda.synchronize()
            t = buf.cpu()
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            buf.copy_(t)
            torch.cuda.synchronize()
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        return buf
    eh.dp.allreduce_partials = _host_allreduce

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
b, nsteps = 4096, 40
spec, theta, X, f, y = util.rbq10_case(world * b * nsteps, "tanh", True, 0.1)
# rank r's shard: its slice of every global batch, so that global step i == samples [i*world*b, (i+1)*world*b)
sel = np.concatenate([np.arange(i * world * b + rank * b, i * world * b + (rank + 1) * b) for i in range(nsteps)])
eng = util.load_engine(spec, theta, X[:, sel], {k: v[sel] for k, v in f.items()}, {k: v[sel] for k, v in y.items()})
eng.opt_init("Adam", 0.01)
# Both ranks' kernels must fit on the one GPU at the same time: a step kernel fills a CU per workgroup, and a rank
# whose 256 workgroups all sit waiting for the peer's sums would keep the peer's kernel from ever being scheduled.
eng.set_option("max_blocks", max(1, 128 // world))
SPEC = os.environ.get("EH_TOOL_SPECIALIZE", "0") == "1"      # step kernels compiled at run time around the descriptor (incl. the cross-GPU one)
drv = eh.dp.DataParallel(eng, fused=True, specialize=SPEC, debug_fail_selftest_rank=os.environ.get("EH_TOOL_FAIL_SELFTEST"))
dist.barrier()
t0 = time.perf_counter()
for i in range(nsteps):
    drv.step(i * b, b)
eng.synchronize(); torch.cuda.synchronize()
dt = time.perf_counter() - t0
th = eng.get_params()
ref = util.load_engine(spec, theta, X, f, y); ref.opt_init("Adam", 0.01)
for i in range(nsteps):
    ref.train_step(i * world * b, world * b, want_loss=False)
err = float(np.max(np.abs(th - ref.get_params())))
# replicas must be bitwise identical: every rank sums the same shards in the same order
t = torch.from_numpy(th.copy()); tl = [torch.empty_like(t) for _ in range(world)]
dist.all_gather(tl, t)
same = all(bool(torch.equal(tl[0], q)) for q in tl)
print(f"rank {rank}: p2p={drv.p2p} p2p_mode={drv.p2p_mode} jit_kernels={eng.jit_status()[0]} steps={nsteps} {1e6 * dt / nsteps:.1f} us/step max|theta-ref|={err:.2e} replicas_identical={same}", flush=True)
ok = err <= 3e-5 and same
if os.environ.get("EH_TOOL_FORCE_TIMEOUT") == "1":
    # recovery: one rank (0, or EH_TOOL_TIMEOUT_RANK) runs one step more than the others, so its exchange never completes and runs into the 2 s deadline; check() must
    # notice on every rank, drop to the all-reduce exchange and re-synchronise parameters AND optimiser state from rank 0
    for i in range(3):
        drv.step(i * b, b)
    if rank == int(os.environ.get("EH_TOOL_TIMEOUT_RANK", "0")) % world:          # (the rank that runs ahead: any one of them)
        drv.step(3 * b, b)
    healthy = drv.check()
    for i in range(12):
        drv.step((4 + i) * b, b)
    eng.synchronize(); torch.cuda.synchronize()
    m_, v_, bt_ = eng.get_opt_state()
    state = torch.from_numpy(np.concatenate([eng.get_params(), m_, v_, bt_]).copy()); sl = [torch.empty_like(state) for _ in range(world)]
    dist.all_gather(sl, state)
    same_state = all(bool(torch.equal(sl[0], q)) for q in sl)
    print(f"rank {rank}: forced timeout: check() -> {healthy}, p2p={drv.p2p}, after recovery + 12 steps theta/m/v/beta identical={same_state} finite={bool(np.isfinite(state.numpy()).all())}", flush=True)
    ok = ok and (not healthy) and (not drv.p2p) and same_state
    eng.close(); ref.close()
    dist.barrier(); dist.destroy_process_group()
    sys.exit(0 if ok else 1)
# timing at the headline batch
B = 65536
spec2, theta2, X2, f2, y2 = util.rbq10_case(8 * B, "tanh", True, 0.0)
e2 = util.load_engine(spec2, theta2, X2, f2, y2); e2.opt_init("Adam", 0.01)
e2.set_option("max_blocks", max(1, 128 // world))
d2 = eh.dp.DataParallel(e2, fused=True, specialize=SPEC)
cal = d2.calibrate(0, B, 100)
if rank == 0: print("calibration:", cal, flush=True)
for i in range(50): d2.step((i % 8) * B, B)
e2.synchronize(); dist.barrier()
t0 = time.perf_counter()
NS = int(os.environ.get("EH_SOAK_STEPS", "500"))
for i in range(NS): d2.step((i % 8) * B, B)
e2.synchronize(); torch.cuda.synchronize()
print(f"rank {rank}: B=65536 per rank, p2p={d2.p2p}: {1e6 * (time.perf_counter() - t0) / NS:.1f} us/step ({world} ranks share one GPU, {max(1, 128 // world)} workgroups each)", flush=True)
t = torch.from_numpy(e2.get_params().copy()); tl = [torch.empty_like(t) for _ in range(world)]
dist.all_gather(tl, t)
same2 = all(bool(torch.equal(tl[0], q)) for q in tl)
print(f"rank {rank}: after {NS} steps replicas_identical={same2} finite={bool(np.isfinite(t.numpy()).all())}", flush=True)
ok = ok and same2
eng.close(); ref.close(); e2.close()
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
