"""Cost of the peer-to-peer machinery itself (ticket, fold, uncached stores, fence, flag, flag wait, uncached
shard reads) with the exchange looped back onto the one GPU: world = 1, the rank publishes to and waits for itself."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import util
B = 65536
spec, theta, X, f, y = util.rbq10_case(8 * B, "tanh", True, 0.0)
SPEC = int(os.environ.get("EH_TOOL_SPECIALIZE", "1"))        # the run-time specialised kernels (what bench.py runs), 0 = built ahead of time
for mode in ("plain fused", "p2p loopback"):
    eng = util.load_engine(spec, theta, X, f, y); eng.opt_init("Adam", 0.01); eng.set_option("fused_update", 1); eng.set_option("specialize", SPEC)
    if mode != "plain fused":
        hd = eng.p2p_init(1, 0); eng.p2p_attach([hd]); assert eng.p2p_selftest(4)
    for i in range(200): eng.dp_fused_step((i % 8) * B, B)
    eng.synchronize()
    t0 = time.perf_counter()
    for i in range(3000): eng.dp_fused_step((i % 8) * B, B)
    eng.synchronize()
    print(f"{mode}: {1e6 * (time.perf_counter() - t0) / 3000:.2f} us/step (kernels compiled at run time: {eng.jit_status()[0]})", flush=True)
    th = eng.get_params(); eng.close()
    if mode == "plain fused": th0 = th
print("max |theta_p2p - theta_plain| =", float(np.max(np.abs(th - th0))))
