"""Where do the slow calls among the first few train() calls of a process go?  The tutorial's small and large model, eight calls each, every
call's TrainResults.timing part by part (and whether a generation-2 collection of Python's GC ran inside it)."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import easyhybrid_jl_amd as eh
from easyhybrid_jl_amd.synthetic import RBQ10_PARAMS, make_synth_rbq10
cols = make_synth_rbq10(5000, seed=42)
gcs = []
gc.callbacks.append(lambda phase, info: gcs.append((phase, info["generation"], time.perf_counter())))
for hidden in ((16, 16), (1024, 512, 256, 128, 64)):
    model = eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, dict(RBQ10_PARAMS), ["rb"], ["Q10"], hidden_layers=list(hidden),
                                    activation="sigmoid", scale_nn_outputs=True, input_batchnorm=True)
    kw = dict(nepochs=20, batchsize=64, opt=eh.RMSProp(0.01), loss_types=["mse", "nse"], keep_history=False)
    for rep in range(8):
        del gcs[:]
        t0 = time.perf_counter(); r = eh.train(model, cols, timing=True, **kw); t1 = time.perf_counter()
        tm = r.timing
        g2 = sum(1 for (ph, gen, _) in gcs if ph == "start" and gen == 2)
        gct = 0.0; st = None
        for ph, gen, t in gcs:
            if ph == "start": st = t
            elif st is not None: gct += t - st; st = None
        print("%-28s call %d: %6.1f ms | " % (str(hidden), rep, 1e3 * (t1 - t0)) + " ".join("%s %.2f" % (k[:-2], 1e3 * tm[k]) for k in ("prepare_s", "engine_s", "upload_s", "setup_s", "initial_eval_s", "loop_s", "steps_s", "eval_s", "host_s", "final_predictions_s"))
              + " | after return %.2f | gc: %d collections (%d gen-2), %.2f ms" % (1e3 * (t1 - t0 - tm["call_s_before_close"]), sum(1 for g in gcs if g[0] == "start"), g2, 1e3 * gct), flush=True)
