"""Diagnostic: phase stamps of the LAST step of a multi-step launch (EH_MODE_TRAIN_MULTI) at batch 64:
   EH_JIT_DEFINES="EH_STAMPS" EH_SPECIALIZE=1 EH_JIT_CACHE=0 EH_NO_AOT_SPEC=1 python tools/stamps_multistep.py
   EH_TOOL_BN=1: the tutorial's model (input BatchNorm + sigmoid) instead of plain tanh"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import easyhybrid_jl_amd as eh
from easyhybrid_jl_amd.synthetic import RBQ10_PARAMS, make_synth_rbq10
model = eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, dict(RBQ10_PARAMS), ["rb"], ["Q10"], hidden_layers=[16, 16], activation="sigmoid" if os.environ.get("EH_TOOL_BN") else "tanh",
                                scale_nn_outputs=True, input_batchnorm=bool(os.environ.get("EH_TOOL_BN")))
cols = make_synth_rbq10(4000, seed=1)
X = np.stack([cols["sw_pot"], cols["dsw_pot"]]).astype(np.float32)
eng = model.engine(0)
eng.set_data(0, X, [cols["ta"]], [cols["reco"]])
eng.set_params(model.initialparameters(1)); eng.opt_init("Adam", 0.01); eng.set_option("fused_update", 1)
eng.loss_and_grad(count=64)
buf = (C.c_uint64 * 32)()
for multi in (0, 1):
    eng.set_option("multi_step", multi)
    eng._lib.eh_debug_stamps(eng._h, buf, 32)
    for k in range(5): eng.train_epoch(64, seed=k, shuffle=True, want_loss=False)
    eng._lib.eh_debug_stamps(eng._h, buf, 32)
    st = np.array(list(buf), dtype=np.int64).reshape(16, 2)
    print("multi_step=%d  jit %s" % (multi, eng.jit_status()[0]))
    order = [i for i in range(16) if st[i, 0] > 0]
    order.sort(key=lambda i: st[i, 0])
    prev = None
    for i in order:
        print("   stamp %2d  +%7d cycles" % (i, 0 if prev is None else st[i, 0] - st[prev, 0]))
        prev = i
eng.close()
