"""train(model, data, distributed=True) with two ranks on ONE GPU (gloo handshake, peer-to-peer gradient exchange):
both ranks must return bitwise-identical TrainResults, and the fit must be as good as single-process training
(the sample order differs -- per-shard shuffle -- so the parameters are close, not equal).

    EH_MAX_BLOCKS=64 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29633 tools/train_two_ranks.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
import easyhybrid_jl_amd as eh

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)


def _host_allreduce(buf, group=None):        # (see tools/p2p_two_ranks.py: gloo's own CUDA path races on this build)
    if buf.is_cuda:
        torch.cuda.synchronize(); t = buf.cpu(); dist.all_reduce(t, group=group); buf.copy_(t); torch.cuda.synchronize()
    else:
        dist.all_reduce(buf, group=group)
    return buf
eh.dp.allreduce_partials = _host_allreduce

ok = True
for bn in (False, True):
    cols = eh.synthetic.make_synth_rbq10(20000, seed=5, nan_frac=0.05)
    if not bn:
        cols = dict(cols); cols["sw_pot"] = cols["sw_pot"] / 50; cols["dsw_pot"] = cols["dsw_pot"] / 50
    model = eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, dict(eh.synthetic.RBQ10_PARAMS), ["rb"], ["Q10"],
                                    hidden_layers=[16, 16], activation="tanh", scale_nn_outputs=True, input_batchnorm=bn)
    kw = dict(nepochs=8, batchsize=1024, opt=eh.Adam(0.01), loss_types=["mse", "r2"], random_seed=11)
    out = eh.train(model, cols, distributed=True, **kw)
    ref = eh.train(model, cols, distributed=False, **kw)
    t = torch.from_numpy(np.concatenate([out.ps, [out.best_loss], out.val_obs_pred["reco_pred"][:100]]).astype(np.float64))
    tl = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(tl, t)
    same = all(bool(torch.equal(tl[0], q)) for q in tl)
    v_d, v_r = out.val_history[-1]["mse"]["sum"], ref.val_history[-1]["mse"]["sum"]
    print(f"rank {rank}: input_batchnorm={bn} val mse distributed {v_d:.5f} vs single-process {v_r:.5f}; "
          f"first-epoch {out.val_history[0]['mse']['sum']:.3f}; results_identical_across_ranks={same}", flush=True)
    ok = ok and same and v_d <= 1.25 * v_r + 1e-3 and v_d < 0.5 * out.val_history[0]["mse"]["sum"]
# a two-target model (flux partitioning: NEE and GPP observed, different gaps): the per-target normalisers of the GLOBAL batch
# go round before every pass (eh_dp_counts + a 12-float all-reduce), per-target losses
rng = np.random.default_rng(3)
n = 16000
cols = {f"x{i}": rng.standard_normal(n).astype(np.float32) for i in range(4)}
cols["SW_IN"] = (rng.random(n) * 400).astype(np.float32); cols["TA"] = (rng.random(n) * 30).astype(np.float32)
gpp = 0.004 * cols["SW_IN"] * (1 + 0.3 * np.tanh(cols["x0"])); reco = (1.5 + 0.5 * np.tanh(cols["x1"])) * 1.6 ** (0.1 * (cols["TA"] - 15))
cols["GPP"] = (gpp + 0.05 * rng.standard_normal(n)).astype(np.float32); cols["NEE"] = (reco - gpp + 0.05 * rng.standard_normal(n)).astype(np.float32)
cols["NEE"][rng.random(n) < 0.3] = np.nan; cols["GPP"][:n // 2][rng.random(n // 2) < 0.6] = np.nan      # the first shard sees few GPP values
model = eh.constructHybridModel([f"x{i}" for i in range(4)], ["SW_IN", "TA"], ["NEE", "GPP"], eh.FluxPartModelQ10,
                                {"RUE": (0.005, 0.0, 0.02), "Rb": (1.5, 0.0, 6.0), "Q10": (1.6, 1.0, 4.0)}, ["RUE", "Rb"], ["Q10"],
                                hidden_layers=[16, 16], activation="tanh", scale_nn_outputs=True)
kw = dict(nepochs=6, batchsize=1024, opt=eh.Adam(0.01), loss_types=["mse", "r2"], random_seed=11, training_loss=eh.PerTarget(("mse", "mae")))
out = eh.train(model, cols, distributed=True, **kw)
ref = eh.train(model, cols, distributed=False, **kw)
t = torch.from_numpy(np.concatenate([out.ps, [out.best_loss], out.val_obs_pred["NEE_pred"][:100]]).astype(np.float64))
tl = [torch.empty_like(t) for _ in range(world)]
dist.all_gather(tl, t)
same = all(bool(torch.equal(tl[0], q)) for q in tl)
v_d, v_r = out.val_history[-1]["mse"]["sum"], ref.val_history[-1]["mse"]["sum"]
print(f"rank {rank}: two targets, PerTarget(mse, mae): val mse distributed {v_d:.5f} vs single-process {v_r:.5f}; "
      f"first-epoch {out.val_history[0]['mse']['sum']:.3f}; results_identical_across_ranks={same}", flush=True)
ok = ok and same and v_d <= 1.25 * v_r + 1e-3 and v_d < 0.5 * out.val_history[0]["mse"]["sum"]
# a two-pass training loss (kgeLoss): the moments of the GLOBAL batch's predictions go round twice ahead of every pass (eh_dp_moments)
cols = eh.synthetic.make_synth_rbq10(12000, seed=9, nan_frac=0.05)
cols = dict(cols); cols["sw_pot"] = cols["sw_pot"] / 50; cols["dsw_pot"] = cols["dsw_pot"] / 50
model = eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, dict(eh.synthetic.RBQ10_PARAMS), ["rb"], ["Q10"],
                                hidden_layers=[16, 16], activation="tanh", scale_nn_outputs=True)
kw = dict(nepochs=5, batchsize=1024, opt=eh.Adam(0.01), loss_types=["kge", "mse"], training_loss="kgeLoss", random_seed=11)
out = eh.train(model, cols, distributed=True, **kw)
ref = eh.train(model, cols, distributed=False, **kw)
t = torch.from_numpy(np.concatenate([out.ps, [out.best_loss], out.val_obs_pred["reco_pred"][:100]]).astype(np.float64))
tl = [torch.empty_like(t) for _ in range(world)]
dist.all_gather(tl, t)
same = all(bool(torch.equal(tl[0], q)) for q in tl)
k_d, k_r, k_0 = out.val_history[-1]["kge"]["sum"], ref.val_history[-1]["kge"]["sum"], out.val_history[0]["kge"]["sum"]
print(f"rank {rank}: kgeLoss: val kge distributed {k_d:.4f} vs single-process {k_r:.4f}; first-epoch {k_0:.4f}; results_identical_across_ranks={same}", flush=True)
ok = ok and same and k_d >= k_r - 0.05 and k_d > k_0 + 0.1
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
