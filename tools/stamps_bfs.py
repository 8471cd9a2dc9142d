"""Diagnostic: phase breakdown of the sample-owned bf16 training kernel's LAST tile (eh_bf16_sample.hpp) from in-kernel stamps
(tools/build_stamps_lib.sh wide 2_8_2; EASYHYBRID_HIP_LIB=.../libeasyhybrid_hip_stamps.so EH_PRECISION=2 python tools/stamps_bfs.py [B])."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import util, test_gpu_parity as tg
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
hidden = tuple(int(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else (128, 128)
spec, theta, X, f, y = tg._rs6_case(int(os.environ.get("EH_P", "32")), hidden, B)
eng = util.load_engine(spec, theta, X, f, y)
eng.opt_init("Adam", 1e-3)
eng.set_option("precision", int(os.environ.get("EH_PRECISION", "2")))
if "EH_SPECIALIZE" in os.environ: eng.set_option("specialize", int(os.environ["EH_SPECIALIZE"]))      # with EH_JIT_DEFINES=EH_STAMPS EH_NO_AOT_SPEC=1: the run-time specialised kernel, stamped
if "EH_VARIANT" in os.environ: eng.set_option("variant", int(os.environ["EH_VARIANT"]))
buf = (C.c_uint64 * 32)()
eng._lib.eh_debug_stamps(eng._h, buf, 32)
for _ in range(50): eng.train_step(0, B, want_loss=False)
eng._lib.eh_debug_stamps(eng._h, buf, 32)
st = np.array(list(buf), dtype=np.int64).reshape(16, 2)
names = ["records -> xb, scratch", "layer 0", "hidden forward", "output layer + sigma", "mechanistic stage", "backward last hidden", "hidden backward (dH, dZ)",
         "staged sets before the last + last stage + barrier", "dW of the last set", "closing barrier", "(loop exit)", "epilogue"]
print(f"last tile: {st[10,0]-st[0,0]} cycles; clock {(st[12,0]-st[0,0])/max(1,(st[12,1]-st[0,1])*10):.2f} GHz")
print(f"   prologue (image)           {st[14,0]-st[13,0]:8d} cycles; whole kernel {st[12,0]-st[13,0]} cycles")
for i, n in enumerate(names):
    print(f"   {n:52s} {st[i+1,0]-st[i,0]:8d} cycles")
if st[15, 0] > 0:
    print(f"   (fine) output-layer MFMAs done + sigma table read at +{st[15,0]-st[3,0]} cycles after the hidden forward; sigma + scratch writes + wave sync {st[4,0]-st[15,0]}")
eng.close()
