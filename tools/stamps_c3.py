"""Diagnostic: phase breakdown of the per-wave step kernel on the BASELINE config 3 shape (build with `make STAMPS_FINE=1`, or EH_JIT_DEFINES="EH_STAMPS EH_STAMPS_FINE" EH_SPECIALIZE=1 EH_JIT_CACHE=0; EH_VARIANT picks the variant)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import easyhybrid_jl_amd as eh
from easyhybrid_jl_amd.synthetic import EXPO2POOL_PARAMS, make_synth_expo2pool
B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
model = eh.constructHybridModel([f"x{i}" for i in range(8)], ["T"], ["Resp_obs"], eh.Expo2Pool, dict(EXPO2POOL_PARAMS),
                                ["R0a", "ka", "R0b", "kb"], [], hidden_layers=[64, 64], activation="tanh", scale_nn_outputs=True)
cols = make_synth_expo2pool(B, 1)
eng = model.engine(0)
eng.set_data(0, np.stack([cols[f"x{i}"] for i in range(8)]), [cols["T"]], [cols["Resp_obs"]])
eng.set_params(model.initialparameters(1)); eng.opt_init("Adam", 0.01)
if os.environ.get("EH_VARIANT"): eng.set_option("variant", int(os.environ["EH_VARIANT"]))
buf = (C.c_uint64 * 32)()
eng._lib.eh_debug_stamps(eng._h, buf, 32)
for _ in range(30): eng.train_step(0, B, want_loss=False)
eng._lib.eh_debug_stamps(eng._h, buf, 32)
st = np.array(list(buf), dtype=np.int64).reshape(16, 2)
names = ["stage weights", "init acc", "load record (last tile)", "layer0", "hidden", "out layer", "mech+loss", "backward", "block reduce", "slab write"]
print(f"B={B}: kernel {st[10,0]-st[0,0]} cycles = {(st[10,1]-st[0,1])*10} ns -> {(st[10,0]-st[0,0])/((st[10,1]-st[0,1])*10):.2f} GHz (workgroup 0, thread 0; tile stamps = its last tile)")
print("   end-of-kernel reduction (cycles): wave sums %d, barrier 1 (waits for the slowest wave) %d, row sums %d, scatter into the wave's region %d, barrier 2 %d" % (
    st[13,0]-st[8,0], st[14,0]-st[13,0], st[11,0]-st[14,0], st[12,0]-st[11,0], st[9,0]-st[12,0]))
for i, nme in enumerate(names):
    j = i + 1
    while j < 10 and st[j, 0] == 0: j += 1
    if st[i, 0] == 0: continue
    print(f"   {nme:24s} {st[j,0]-st[i,0]:8d} cycles {(st[j,1]-st[i,1])*10:8d} ns")
eng.close()
