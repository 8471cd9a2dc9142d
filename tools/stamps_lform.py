"""Diagnostic: phase stamps of the layer-wise form's few-rows chain kernel (eh_lform_tailchain_kernel), workgroup 0, on the reference's
GPU tutorial network at batch 64 (or argv[1]):
   bash tools/stamps_lform.sh && EASYHYBRID_HIP_LIB=easyhybrid.jl_amd/libeasyhybrid_hip_stamps.so python tools/stamps_lform.py [batch]
slots: 0 entry | 1 kernel arguments in | 2 inputs / biases staged | 3 + l forward layer l of the suffix | 8 mechanistic stage | 9 + j
delta products from the top | 15 end"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import easyhybrid_jl_amd as eh
from easyhybrid_jl_amd.synthetic import RBQ10_PARAMS, make_synth_rbq10
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
model = eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, dict(RBQ10_PARAMS), ["rb"], ["Q10"], hidden_layers=[1024, 512, 256, 128, 64],
                                activation="sigmoid", scale_nn_outputs=True, input_batchnorm=True)
cols = make_synth_rbq10(8 * B, seed=1)
X = np.stack([cols["sw_pot"], cols["dsw_pot"]]).astype(np.float32)
eng = model.engine(0)
eng.set_data(0, X, [cols["ta"]], [cols["reco"]])
eng.set_params(model.initialparameters(1)); eng.opt_init("RMSProp", 0.001)
buf = (C.c_uint64 * 32)()
eng._lib.eh_debug_stamps(eng._h, buf, 32)
for rep in range(3):
    for s in range(20): eng.train_step((s % 8) * B, B, want_loss=False)
    eng.synchronize()
    eng._lib.eh_debug_stamps(eng._h, buf, 32)
    st = np.array(list(buf), dtype=np.int64).reshape(16, 2)
    order = sorted([i for i in range(16) if st[i, 0] > 0], key=lambda i: st[i, 0])
    print("batch %d, run %d: total %d cycles, %.2f us wall" % (B, rep, st[order[-1], 0] - st[order[0], 0], (st[order[-1], 1] - st[order[0], 1]) / 100.0))
    if os.environ.get("EH_STAMP_DW") and st[14, 0] > 0 and st[15, 0] > 0:
        order = [i for i in order if i < 14]
        print("   launch span (first workgroup start -> last workgroup's thread 0 end): %.2f us; this workgroup starts %.2f us after the first, ends %.2f us before the last"
              % ((st[15, 0] - st[14, 0]) / 100.0, (st[order[0], 1] - st[14, 0]) / 100.0, (st[15, 0] - st[order[-1], 1]) / 100.0))
    prev = None
    for i in order:
        print("   stamp %2d  +%7d cycles" % (i, 0 if prev is None else st[i, 0] - st[prev, 0]))
        prev = i
eng.close()
