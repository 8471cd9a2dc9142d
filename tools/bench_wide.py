"""Step-time probe for the BASELINE.json configs[4] shape: MLP [32,128,128,6] + Rs_components, batch B.
python tools/bench_wide.py [B] [steps]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import easyhybrid_jl_amd as eh
from oracle import hybrid_oracle as ho
from tests import util, test_gpu_parity as tg

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
hidden = tuple(int(v) for v in sys.argv[3].split(",")) if len(sys.argv) > 3 else (128, 128)
import os
P = int(os.environ.get("EH_P", "32"))
spec, theta, X, f, y = tg._rs6_case(P, hidden, B)
eng = util.load_engine(spec, theta, X, f, y)
if "EH_VARIANT" in os.environ:
    eng.set_option("variant", int(os.environ["EH_VARIANT"]))
if "EH_ROW_SPLIT" in os.environ:
    eng.set_option("row_split", int(os.environ["EH_ROW_SPLIT"]))
eng.opt_init("Adam", 1e-3)
for _ in range(20):
    eng.train_step(0, B, want_loss=False)
eng.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    eng.train_step(0, B, want_loss=False)
eng.synchronize()
dt = (time.perf_counter() - t0) / steps
dims = [P, *hidden, 6]
flops = 6 * B * sum(a * b for a, b in zip(dims[:-1], dims[1:]))
print(f"P={P} variant={os.environ.get('EH_VARIANT', 'default')} row_split={os.environ.get('EH_ROW_SPLIT', 'default')} hidden={hidden} B={B}: {dt*1e6:.1f} us/step, {B/dt/1e9:.3f} G samples/s, {flops/dt/1e12:.1f} TFLOP/s (fwd+bwd, 6*B*sum(in*out))")
