"""The primitives the weak-scaling prediction of DESIGN.md section 4 is built from, measured on ONE GPU at the headline batch
(65 536 samples per rank, run-time specialised kernels):
  plain fused step | the peer-to-peer machinery looped back onto the one GPU (world = 1) | the library's RCCL communicator at
  world = 1 around the fused step and around the two-kernel step (the launch + completion cost of an RCCL all-reduce of the
  8 x 340 / 340 floats, without any link latency -- a lower bound of what a node pays)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from easyhybrid_jl_amd.engine import HybridEngine
from tests import util
B, N = 65536, 3000
spec, theta, X, f, y = util.rbq10_case(8 * B, "tanh", True, 0.0)
out = {}
for mode in ("plain fused", "p2p loopback", "p2p loopback, published from the next prologue", "rccl world=1 fused", "two-kernel", "rccl world=1 two-kernel"):
    eng = util.load_engine(spec, theta, X, f, y); eng.opt_init("Adam", 0.01)
    fused = "two-kernel" not in mode
    eng.set_option("fused_update", int(fused)); eng.set_option("specialize", 1)
    if mode.startswith("p2p loopback"):
        hd = eng.p2p_init(1, 0); eng.p2p_attach([hd]); assert eng.p2p_selftest(4)
        if "prologue" in mode: eng.set_option("p2p_mode", 1)
    if mode.startswith("rccl"):
        eng.comm_init(HybridEngine.comm_unique_id(), 1, 0)
    if mode.startswith("rccl"):
        step = lambda i: eng.dp_train_step((i % 8) * B, B)
    elif fused:
        step = lambda i: eng.dp_fused_step((i % 8) * B, B)
    else:
        step = lambda i: eng.train_step((i % 8) * B, B, want_loss=False)
    for i in range(200): step(i)
    eng.synchronize()
    t0 = time.perf_counter()
    for i in range(N): step(i)
    eng.synchronize()
    out[mode] = 1e6 * (time.perf_counter() - t0) / N
    print(f"{mode}: {out[mode]:.2f} us/step", flush=True)
    eng.close()
print(f"peer-to-peer machinery (loop-back): +{out['p2p loopback'] - out['plain fused']:.2f} us with the election in the step's epilogue, "
      f"+{out['p2p loopback, published from the next prologue'] - out['plain fused']:.2f} us published from the next kernel's prologue;  RCCL all-reduce at world 1: "
      f"+{out['rccl world=1 fused'] - out['plain fused']:.2f} us (fused), +{out['rccl world=1 two-kernel'] - out['two-kernel']:.2f} us (two-kernel)")
