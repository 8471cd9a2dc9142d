import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np
import easyhybrid_jl_amd as eh
from easyhybrid_jl_amd.synthetic import RBQ10_PARAMS, make_synth_rbq10
cols = make_synth_rbq10(5000, seed=42)
model = eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, dict(RBQ10_PARAMS), ["rb"], ["Q10"],
                                hidden_layers=[1024, 512, 256, 128, 64], activation="sigmoid", scale_nn_outputs=True, input_batchnorm=True)
X = np.stack([cols["sw_pot"], cols["dsw_pot"]]).astype(np.float32)
eng = model.engine(0)
eng.set_data(0, X, [cols["ta"]], [cols["reco"]])
eng.set_params(model.initialparameters(1)); eng.opt_init("RMSProp", 0.01)
def t(fn, n):
    fn(); eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    eng.synchronize()
    return (time.perf_counter() - t0) / n
steps = lambda: [eng.train_step((s % 70) * 64, 64, want_loss=False) for s in range(63)]
print("63 train_step calls, contiguous: %.1f us/step" % (1e6 * t(steps, 10) / 63))
for sh in (False, True):
    print("train_epoch(64, shuffle=%s): %.1f us/step" % (sh, 1e6 * t(lambda: eng.train_epoch(64, seed=3, shuffle=sh, want_loss=False), 10) / 63))
perm = np.random.default_rng(0).permutation(4000).astype(np.int32)
stepsg = lambda: [eng.train_step(s * 64, 64 if s < 62 else 32, want_loss=False, idx=perm) for s in range(63)]
print("63 train_step calls, host idx: %.1f us/step" % (1e6 * t(stepsg, 10) / 63))
eng.close()
# what train() does per epoch: train_epoch, then the two evaluation passes
eng = model.engine(0)
ntr = 4000
eng.set_data(0, X[:, :ntr], [cols["ta"][:ntr]], [cols["reco"][:ntr]])
eng.set_data(1, X[:, ntr:], [cols["ta"][ntr:]], [cols["reco"][ntr:]])
eng.set_params(model.initialparameters(1)); eng.opt_init("RMSProp", 0.01)
for rep in range(2):
    te = tv = 0.0
    for ep in range(10):
        t0 = time.perf_counter(); eng.train_epoch(64, seed=ep, shuffle=True, want_loss=False); eng.synchronize(); t1 = time.perf_counter()
        eng.eval(0); eng.eval(1); t2 = time.perf_counter()
        te += t1 - t0; tv += t2 - t1
    print("epoch of 63 steps (62 x 64 + 32): %.1f us/step; two evaluation passes: %.2f ms" % (1e6 * te / 10 / 63, 1e3 * tv / 10))
eng.close()
