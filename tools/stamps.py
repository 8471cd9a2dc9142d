"""Diagnostic: phase breakdown of the fused step kernel from in-kernel stamps.  Either build the library with `make STAMPS=1`,
or stamp the run-time specialised kernel of a normal build:  EH_JIT_DEFINES="EH_STAMPS" EH_SPECIALIZE=1 EH_JIT_CACHE=0 python tools/stamps.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import easyhybrid_jl_amd as eh
from easyhybrid_jl_amd.synthetic import RBQ10_PARAMS, make_synth_rbq10
model = eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, dict(RBQ10_PARAMS), ["rb"], ["Q10"],
                                hidden_layers=[16, 16], activation="tanh", scale_nn_outputs=True)
cols = make_synth_rbq10(1 << 18, seed=1)
X = np.stack([cols["sw_pot"], cols["dsw_pot"]]).astype(np.float32)
eng = model.engine(0)
eng.set_data(0, X, [cols["ta"]], [cols["reco"]])
eng.set_params(model.initialparameters(1)); eng.opt_init("Adam", 0.01)
eng.set_option("fused_update", int(os.environ.get("EH_FUSED", "0")))
eng.set_option("variant", int(os.environ.get("EH_VARIANT", "0")))
buf = (C.c_uint64 * 32)()
eng._lib.eh_debug_stamps(eng._h, buf, 32)
names = ["stage weights", "init acc", "load record", "layer0", "hidden", "out layer", "mech+loss", "backward", "block reduce", "slab write"]
for B in (64, 65536, 131072):
    for _ in range(200): eng.train_step(0, B, want_loss=False)
    eng._lib.eh_debug_stamps(eng._h, buf, 32)
    st = np.array(list(buf), dtype=np.int64).reshape(16, 2)
    print(f"B={B}: total {st[10,0]-st[0,0]} cycles = {(st[10,1]-st[0,1])*10} ns -> clock {(st[10,0]-st[0,0])/((st[10,1]-st[0,1])*10):.2f} GHz")
    print("   reduce detail (cycles): pre-barrier1 %d, barrier1 %d, cross-lane sums %d, lds writes %d, barrier2 %d" % (
        st[13,0]-st[8,0], st[14,0]-st[13,0], st[11,0]-st[14,0], st[12,0]-st[11,0], st[9,0]-st[12,0]))
    for i, nme in enumerate(names):
        j = i + 1
        while j < 10 and st[j, 0] == 0: j += 1
        if st[i, 0] == 0: continue
        print(f"   {nme:14s} {st[j,0]-st[i,0]:8d} cycles  {(st[j,1]-st[i,1])*10:8d} ns")
eng.close()
