"""where eh_create / eh_destroy spend their time (EH_CREATE_TRACE=1 prints the checkpoints of the second create)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import easyhybrid_jl_amd as eh
from easyhybrid_jl_amd.synthetic import RBQ10_PARAMS
for bn in (False, True):
    model = eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, dict(RBQ10_PARAMS), ["rb"], ["Q10"], hidden_layers=[16, 16], activation="tanh", scale_nn_outputs=True, input_batchnorm=bn)
    model.engine(0).close()
    os.environ["EH_CREATE_TRACE"] = "1"
    t0 = time.perf_counter(); e = model.engine(0); t1 = time.perf_counter(); e.close(); t2 = time.perf_counter()
    del os.environ["EH_CREATE_TRACE"]
    print("input_batchnorm=%s: engine() %.2f ms, close() %.2f ms" % (bn, 1e3 * (t1 - t0), 1e3 * (t2 - t1)))
