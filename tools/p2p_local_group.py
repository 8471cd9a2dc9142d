"""ONE process, `world` engines on device 0 (all a one-GPU box allows; on the node: one engine per device), fused step kernels that
exchange their sums themselves through plain pointers to each other's receive buffers (eh_p2p_init_local): the 8-rank walk of the
peer-to-peer protocol that the process limit of the GPU boxes forbids as 8 processes -- every slot of recv[3][EH_GSHARDS][n_acc]
in use, EH_GSHARDS = 8 both the shard count of the float atomics and the peer-slot count.

    python tools/p2p_local_group.py 8            # 2 / 4 / 8 engines

Checks: parameters equal to ONE engine stepping on the union of the windows, replicas bitwise identical, eh_p2p_check_local healthy;
then a forced missed exchange (engine 0 steps alone -> the 2 s deadline) and the recovery: every member back on the local group's
all-reduce with member 0's parameters and optimiser state, bitwise identical after twelve further steps; and the refusal path
(EH_DEBUG_P2P_FAIL_SELFTEST: the group keeps all-reducing)."""
import os
import sys

# eight streams whose kernels wait for each other must not share a hardware queue (the default is four queues per process: a
# kernel that spins until a peer's sums arrive would sit IN FRONT of that peer's kernel).  On the node every device has its own.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import easyhybrid_jl_amd as eh
from easyhybrid_jl_amd import dp
from easyhybrid_jl_amd.engine import HybridEngine
from tests import util

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B, nsteps = 8192, 24
spec, theta, X, f, y = util.rbq10_case(B, "tanh", True, 0.0)
y["reco"][: B // world][::2] = np.nan                      # all the gaps in rank 0's shard: per-shard means would be wrong
per = B // world
win = per // 4


def shard_engines():
    engs = []
    for r in range(world):
        lo, hi = dp.shard_range(B, r, world)
        e = util.load_engine(spec, theta, X[:, lo:hi], {k: v[lo:hi] for k, v in f.items()}, {k: v[lo:hi] for k, v in y.items()})
        e.opt_init("Adam", 0.01)
        e.set_option("fused_update", 1)
        # every member's workgroups must be resident together on the ONE device: a kernel whose workgroups all wait for a peer
        # would otherwise keep that peer's kernel from being scheduled
        e.set_option("max_blocks", max(1, 128 // world))
        engs.append(e)
    HybridEngine.comm_init_local(engs)
    return engs


def union_idx(a):
    return np.concatenate([np.arange(r * per + a, r * per + a + win) for r in range(world)]).astype(np.int32)


def identical(engs, state=False):
    packs = []
    for e in engs:
        p = [e.get_params()]
        if state:
            m, v, bt = e.get_opt_state()
            p += [m, v, np.asarray(bt, np.float32)]
        packs.append(np.concatenate(p))
    return all(np.array_equal(packs[0], q) for q in packs[1:]), bool(np.isfinite(packs[0]).all())


ok = True
engs = shard_engines()
p2p = HybridEngine.p2p_init_local(engs)
P2P_MODE = int(os.environ.get("EH_TOOL_P2P_MODE", "0"))      # 1: a step's sums are published by workgroup 0 of the NEXT kernel on the stream (no election)
if p2p and P2P_MODE:
    for e in engs:
        e.set_option("p2p_mode", P2P_MODE)
ref = util.load_engine(spec, theta, X, f, y); ref.opt_init("Adam", 0.01)
for s in range(nsteps):
    a = (s % 4) * win
    HybridEngine.dp_train_step_group(engs, [a] * world, win)
    ref.train_step(0, world * win, want_loss=False, idx=union_idx(a))
healthy = HybridEngine.p2p_check_local(engs)
same, finite = identical(engs)
err = float(np.max(np.abs(engs[0].get_params() - ref.get_params())))
print(f"local group of {world}: p2p={p2p} mode={P2P_MODE} steps={nsteps} healthy={healthy} max|theta-ref|={err:.2e} replicas_identical={same}", flush=True)
ok = ok and p2p and healthy and same and finite and err <= 2e-5

# a missed exchange: member 0 steps alone, so the sums it waits for never come (2 s deadline); the check must notice, drop every
# member to the local group's all-reduce and hand member 0's parameters and optimiser state to all
engs[0].dp_fused_step(0, win)
healthy = HybridEngine.p2p_check_local(engs)
for s in range(12):
    HybridEngine.dp_train_step_group(engs, [(s % 4) * win] * world, win)
for e in engs:
    e.synchronize()
same_state, finite = identical(engs, state=True)
refused = False
try:
    HybridEngine.p2p_check_local(engs)                      # the group is gone
except eh.EngineError:
    refused = True
print(f"local group of {world}: forced timeout: check -> {healthy}; after recovery + 12 steps through the local all-reduce theta/m/v/beta "
      f"identical={same_state} finite={finite}; a second check is refused={refused}", flush=True)
ok = ok and (not healthy) and same_state and finite and refused
for e in engs:
    e.close()

# the refusal path: a failed self-test leaves every member on the all-reduce, nothing mapped
os.environ["EH_DEBUG_P2P_FAIL_SELFTEST"] = "1"
engs = shard_engines()
p2p = HybridEngine.p2p_init_local(engs)
del os.environ["EH_DEBUG_P2P_FAIL_SELFTEST"]
ref2 = util.load_engine(spec, theta, X, f, y); ref2.opt_init("Adam", 0.01)
for s in range(6):
    a = (s % 4) * win
    HybridEngine.dp_train_step_group(engs, [a] * world, win)
    ref2.train_step(0, world * win, want_loss=False, idx=union_idx(a))
for e in engs:
    e.synchronize()
same, finite = identical(engs)
err = float(np.max(np.abs(engs[0].get_params() - ref2.get_params())))
print(f"local group of {world}: failed self-test: p2p={p2p}; all-reduce instead: max|theta-ref|={err:.2e} replicas_identical={same}", flush=True)
ok = ok and (not p2p) and same and finite and err <= 2e-6
for e in engs:
    e.close()
ref.close(); ref2.close()
sys.exit(0 if ok else 1)
