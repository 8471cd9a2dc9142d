"""Diagnostic: the phases of the fused step kernel (in-kernel stamps) with the exchange machinery off, looped back with the ticket
election (mode 0) and looped back with the publication from the next prologue (mode 1) -- what of the loop-back's extra time is where.
    EH_JIT_DEFINES="EH_STAMPS" EH_JIT_CACHE=0 python tools/stamps_p2p.py
    EH_JIT_DEFINES="EH_STAMPS EH_STAMPS_PROLOGUE" EH_JIT_CACHE=0 python tools/stamps_p2p.py       (the prologue taken apart)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import util
B = 65536
spec, theta, X, f, y = util.rbq10_case(8 * B, "tanh", True, 0.0)
PRO = "EH_STAMPS_PROLOGUE" in os.environ.get("EH_JIT_DEFINES", "")
names = ["prologue (update + image)", "init acc", "load record", "layer0", "hidden", "out layer", "mech+loss", "backward", "block reduce", "accumulate + publish"]
for mode in ("plain", "p2p mode 0", "p2p mode 1"):
    eng = util.load_engine(spec, theta, X, f, y); eng.opt_init("Adam", 0.01)
    eng.set_option("fused_update", 1); eng.set_option("specialize", 1)
    if mode.startswith("p2p"):
        hd = eng.p2p_init(1, 0); eng.p2p_attach([hd])
        if mode.endswith("1"): eng.set_option("p2p_mode", 1)
    buf = (C.c_uint64 * 32)()
    eng._lib.eh_debug_stamps(eng._h, buf, 32)
    for i in range(300): eng.dp_fused_step((i % 8) * B, B)
    eng.synchronize()
    eng._lib.eh_debug_stamps(eng._h, buf, 32)
    st = np.array(list(buf), dtype=np.int64).reshape(16, 2)
    print(f"{mode}: total {st[10,0]-st[0,0]} cycles = {(st[10,1]-st[0,1])*10} ns")
    if PRO:
        seq = [0, 2, 3, 4, 5, 6, 7, 1]
        lab = ["state loads issued", "exchange words requested", "image staged", "exchange words in, summed over ranks", "statistics + X images cleared", "update applied", "scalars, next shards zeroed, barrier"]
        if mode.startswith("p2p") and st[11, 0] and st[12, 0]:      # (slot 13 belongs to the BatchNorm block's stamps further down)
            print("   (exchange words in: the wait loop %d cycles, sums over the ranks in registers %d, barrier + the five scalars out of LDS %d)" % (
                st[11, 0] - st[4, 0], st[12, 0] - st[11, 0], st[5, 0] - st[12, 0]))
        for k in range(len(seq) - 1):
            print(f"   {lab[k]:40s} {st[seq[k+1],0]-st[seq[k],0]:8d} cycles  {(st[seq[k+1],1]-st[seq[k],1])*10:8d} ns", flush=True)
    for i, nme in enumerate(names):
        if PRO and 1 <= i <= 7: continue
        j = i + 1
        while j < 10 and st[j, 0] == 0: j += 1
        if st[i, 0] == 0: continue
        print(f"   {nme:28s} {st[j,0]-st[i,0]:8d} cycles  {(st[j,1]-st[i,1])*10:8d} ns", flush=True)
    eng.close()
