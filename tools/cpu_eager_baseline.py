"""PyTorch-CPU eager autograd + Adam on the same RbQ10 batch: the structurally closest stand-in for
the reference's Lux + Zygote step that can run on the box (BLAS GEMMs, un-fused broadcasts, tape,
boolean-mask gather).  Quoted in DESIGN.md next to the plain-C port; not part of bench.py's JSON."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import hybrid_oracle as ho, torch_twin as tt
B = 65536
spec = ho.rbq10_spec((16, 16), "tanh", True)
X, f, y = ho.make_synth_rbq10(B, 42)
theta = ho.init_theta(spec, 1, np.float32)
res = {}
for nt in (8, 32, 64):
    s = tt.train_step_timed(spec, theta, X, f, y, 10, threads=nt)
    res[nt] = {"ms_per_step": 1e3 * s, "samples_per_s": B / s}
print(json.dumps({"torch_eager_cpu": res, "torch_threads_available": torch.get_num_threads()}))
