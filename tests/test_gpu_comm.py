"""-m gpu: the RCCL communicator inside the library (eh_comm_* / eh_dp_allreduce / eh_dp_train_step), through the C ABI, with
world = 1 -- all a one-GPU box allows (RCCL refuses two ranks on one device); the multi-rank arithmetic of the seam is covered on
the CPU by tests/test_dp_gloo.py and by the virtual-shard tests of test_gpu_parity.py."""
import numpy as np
import pytest

import easyhybrid_jl_amd as eh
from easyhybrid_jl_amd import _lib as L
from easyhybrid_jl_amd.engine import HybridEngine
from oracle import hybrid_oracle as ho
from tests import util

pytestmark = pytest.mark.gpu


def _engine(B=4096, bn=False):
    spec, theta, X, f, y = util.rbq10_case(B, "tanh", True, 0.1)
    if bn:
        spec.input_batchnorm = True
    eng = util.load_engine(spec, theta, X, f, y)
    eng.opt_init("Adam", 0.01)
    return eng, spec, theta, X, f, y


@pytest.mark.parametrize("fused", [0, 1])
def test_world_of_one_trains_like_the_plain_step(fused):
    eng, spec, theta, X, f, y = _engine()
    ref, *_ = _engine()
    uid = HybridEngine.comm_unique_id()
    assert len(uid) == L.EH_COMM_ID_BYTES
    eng.comm_init(uid, 1, 0)
    eng.set_option("fused_update", fused)
    ref.set_option("fused_update", fused)
    for s in range(6):
        first = (s % 4) * 1024
        eng.dp_train_step(first, 1024)
        ref.train_step(first, 1024, want_loss=False)
    a, b = eng.get_params(), ref.get_params()
    assert np.max(np.abs(a - b)) <= 2e-6, np.max(np.abs(a - b))
    if not fused:
        l1, l2 = eng.dp_train_step(0, 1024, want_loss=True), ref.train_step(0, 1024)
        assert abs(l1 - l2) <= 1e-6 * abs(l2)
    eng.comm_destroy()
    eng.close(); ref.close()


def test_allreduce_needs_a_communicator_and_a_valid_buffer():
    eng, *_ = _engine()
    with pytest.raises(eh.EngineError):
        eng.dp_allreduce(L.EH_BUF_GRAD)                    # no eh_comm_init yet
    eng.comm_init(HybridEngine.comm_unique_id(), 1, 0)
    with pytest.raises(eh.EngineError):
        eng.comm_init(HybridEngine.comm_unique_id(), 1, 0)  # already has one
    with pytest.raises(ValueError):
        eng.dp_allreduce(L.EH_BUF_THETA)                    # not a reducible buffer
    with pytest.raises(eh.EngineError):
        eng.dp_allreduce(L.EH_BUF_BNSTAT)                   # no input BatchNorm in this model
    eng.dp_grad(0, 512)
    eng.dp_allreduce(L.EH_BUF_GRAD)                         # in place, in stream order
    eng.dp_apply()
    eng.close()


def test_batchnorm_statistics_go_through_the_library_collective():
    eng, spec, theta, X, f, y = _engine(bn=True)
    ref, *_ = _engine(bn=True)
    eng.set_bn_shift(np.zeros(2, np.float32))
    eng.comm_init(HybridEngine.comm_unique_id(), 1, 0)
    for s in range(3):
        eng.dp_train_step(s * 1024, 1024)
        ref.train_step(s * 1024, 1024, want_loss=False)
    assert np.max(np.abs(eng.get_params() - ref.get_params())) <= 5e-6
    for a, b in zip(eng.get_bn_state(), ref.get_bn_state()):
        assert np.allclose(a, b, rtol=2e-6, atol=1e-7)
    eng.close(); ref.close()


@pytest.mark.parametrize("fused", [0, 1])
def test_multi_target_model_through_the_library_collective(fused):
    """eh_dp_train_step on a two-target model: counts -> all-reduce (EH_BUF_TCOUNT) -> gradient sums with the global weights ->
    all-reduce -> apply; in fused_update mode the seam keeps to that three-kernel path (the weights need the global counts)"""
    from oracle import hybrid_oracle as ho
    rng = np.random.default_rng(31)
    B = 2048
    pars = {"RUE": (0.1, 0.0, 1.0), "Rb": (1.0, 0.0, 6.0), "Q10": (1.5, 1.0, 4.0)}
    spec = ho.HybridSpec(4, [16, 8], "fluxpart", pars, ["RUE", "Rb"], ["Q10"], ["NEE", "GPP"], "tanh", True)
    X = rng.standard_normal((4, B)).astype(np.float32)
    f = {"SW_IN": (rng.random(B) * 400).astype(np.float32), "TA": (rng.random(B) * 30).astype(np.float32)}
    y = {"NEE": (5 + rng.standard_normal(B)).astype(np.float32), "GPP": (rng.random(B) * 3).astype(np.float32)}
    y["NEE"][rng.random(B) < 0.3] = np.nan
    theta = ho.init_theta(spec, 6, np.float32)
    eng = util.load_engine(spec, theta, X, f, y); eng.opt_init("Adam", 0.01); eng.set_training_loss(("nseLoss", "mae"))
    ref = util.load_engine(spec, theta, X, f, y); ref.opt_init("Adam", 0.01); ref.set_training_loss(("nseLoss", "mae"))
    eng.comm_init(HybridEngine.comm_unique_id(), 1, 0)
    eng.set_option("fused_update", fused)
    for s in range(4):
        eng.dp_train_step(s * 512, 512)
        ref.train_step(s * 512, 512, want_loss=False)
    assert np.max(np.abs(eng.get_params() - ref.get_params())) <= 3e-6
    eng.comm_destroy()
    eng.close(); ref.close()


# ---- ONE process, several handles: the local group (eh_comm_init_local / eh_dp_train_step_group) ---------------------------------
def _shard_engines(spec, theta, X, f, y, world, opt=("Adam", 0.01)):
    from easyhybrid_jl_amd import dp
    N = X.shape[1]
    engs = []
    for r in range(world):
        lo, hi = dp.shard_range(N, r, world)
        e = util.load_engine(spec, theta, X[:, lo:hi], {k: v[lo:hi] for k, v in f.items()}, {k: v[lo:hi] for k, v in y.items()})
        e.opt_init(*opt)
        engs.append(e)
    return engs


@pytest.mark.parametrize("fused", [0, 1])
@pytest.mark.parametrize("world", [2, 4, 8])
def test_two_handles_of_one_process_train_like_one_engine_on_the_union(world, fused):
    """two / four / eight (= EH_GSHARDS, the most a node has) handles on device 0 driven by this one thread, each holding a shard with
    its own share of missing targets;
    the library's local collective carries the raw sums.  Against ONE engine stepping on the union of the windows."""
    B = 4096
    spec, theta, X, f, y = util.rbq10_case(B, "tanh", True, 0.0)
    y["reco"][: B // world][::2] = np.nan                      # all the gaps in rank 0's shard: per-shard means would be wrong
    engs = _shard_engines(spec, theta, X, f, y, world)
    HybridEngine.comm_init_local(engs)
    for e in engs:
        e.set_option("fused_update", fused)
    ref = util.load_engine(spec, theta, X, f, y); ref.opt_init("Adam", 0.01)
    per = B // world
    win = per // 4
    for s in range(6):
        a = (s % 4) * win
        HybridEngine.dp_train_step_group(engs, [a] * world, win)
        idx = np.concatenate([np.arange(r * per + a, r * per + a + win) for r in range(world)]).astype(np.int32)
        ref.train_step(0, idx.size, want_loss=False, idx=idx)
    th = [e.get_params() for e in engs]
    for t in th[1:]:
        assert np.array_equal(th[0], t), "replicas of a local group must stay bitwise identical"
    assert np.max(np.abs(th[0] - ref.get_params())) <= 2e-6
    if not fused:
        loss = HybridEngine.dp_train_step_group(engs, [0] * world, win, want_loss=True)
        idx = np.concatenate([np.arange(r * per, r * per + win) for r in range(world)]).astype(np.int32)
        assert loss == pytest.approx(ref.train_step(0, idx.size, idx=idx), rel=2e-6)
    engs[1].comm_destroy()                                      # dissolves the whole group
    with pytest.raises(eh.EngineError):
        HybridEngine.dp_train_step_group(engs, [0] * world, win)
    for e in engs:
        e.close()
    ref.close()


def test_local_group_allreduce_needs_the_bracket_and_every_member():
    spec, theta, X, f, y = util.rbq10_case(2048, "tanh", True, 0.1)
    engs = _shard_engines(spec, theta, X, f, y, 2)
    HybridEngine.comm_init_local(engs)
    with pytest.raises(eh.EngineError):
        HybridEngine.comm_init_local(engs)                      # already in a group
    for e in engs:
        e.dp_grad(0, 512)
    with pytest.raises(eh.EngineError):
        engs[0].dp_allreduce(L.EH_BUF_GRAD)                     # outside a bracket
    HybridEngine.comm_group_begin()
    engs[0].dp_allreduce(L.EH_BUF_GRAD)
    with pytest.raises(eh.EngineError):
        HybridEngine.comm_group_end()                           # rank 1 never asked
    HybridEngine.comm_group_begin()
    for e in engs:
        e.dp_allreduce(L.EH_BUF_GRAD)
    HybridEngine.comm_group_end()
    for e in engs:
        e.dp_apply()
    assert np.array_equal(engs[0].get_params(), engs[1].get_params())
    for e in engs:
        e.close()


@pytest.mark.parametrize("world", [2, 8])
@pytest.mark.parametrize("hidden", [(24, 12), (160, 96, 48, 24)])
def test_multi_target_model_under_the_local_group_fused_and_layerwise_form(hidden, world):
    """T = 2 with very different gaps per shard (the per-target weights are those of the GLOBAL batch: eh_dp_counts ahead of the
    pass), on a fused shape and on a shape only the layer-wise form holds (hidden width > 128: advisor finding of round 2 -- the
    layer-wise step re-counted per shard and overwrote the global weights)."""
    rng = np.random.default_rng(8)
    B = 2048
    pars = {"RUE": (0.1, 0.0, 1.0), "Rb": (1.0, 0.0, 6.0), "Q10": (1.5, 1.0, 4.0)}
    spec = ho.HybridSpec(6, list(hidden), "fluxpart", pars, ["RUE", "Rb"], ["Q10"], ["NEE", "GPP"], "tanh", True)
    X = rng.standard_normal((6, B)).astype(np.float32)
    f = {"SW_IN": (rng.random(B) * 400).astype(np.float32), "TA": (rng.random(B) * 30).astype(np.float32)}
    y = {"NEE": rng.standard_normal(B).astype(np.float32), "GPP": (rng.random(B) * 3).astype(np.float32)}
    y["NEE"][: B // 2][rng.random(B // 2) < 0.7] = np.nan      # shard 0 misses most NEE, shard 1 a third of GPP
    y["GPP"][B // 2:][rng.random(B // 2) < 0.3] = np.nan
    theta = ho.init_theta(spec, 3, np.float32)
    engs = _shard_engines(spec, theta, X, f, y, world, opt=("Descent", 0.05))
    HybridEngine.comm_init_local(engs)
    # the global shift of the shifted target sums (what DataParallel all-reduces once): the same vector on every member
    shift = [float(np.nanmean(y[t])) for t in spec.targets]
    for e in engs:
        e.set_target_shift(shift)
    loss = HybridEngine.dp_train_step_group(engs, [0] * world, B // world, want_loss=True)
    l0, g0, _ = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y)
    assert abs(loss - l0) <= 1e-5 * abs(l0), (loss, l0)
    step = (theta.astype(np.float64) - engs[0].get_params().astype(np.float64)) / 0.05          # Descent: theta - lr * grad
    assert util.relerr(step, g0) <= 2e-5, util.relerr(step, g0)
    for e in engs[1:]:
        assert np.array_equal(engs[0].get_params(), e.get_params())
    for e in engs:
        e.close()


@pytest.mark.parametrize("kind", ["kgeLoss", "pearsonLoss"])
def test_two_pass_loss_with_input_batchnorm_under_the_local_group(kind):
    """input BatchNorm AND a two-pass training loss under data parallelism (advisor, round 4): the global statistics of eh_dp_bn_stats serve
    the two forward-only moment passes and the training pass behind them -- stage 0 used to consume them and every later pass of the step
    returned EH_ESTATE.  Two members against ONE engine stepping on the union of their windows."""
    world, B = 2, 2048
    spec, theta, X, f, y = util.rbq10_case(B, "tanh", True, 0.05, hidden=(24, 12))
    spec.input_batchnorm = True
    X = (X * np.float32(2.0) + np.float32(0.5)).astype(np.float32)
    y["reco"][: B // world][::3] = np.nan
    engs = _shard_engines(spec, theta, X, f, y, world, opt=("Descent", 0.05))
    tshift = [float(np.nanmean(y[t])) for t in spec.targets]
    xshift = X.mean(axis=1).astype(np.float32)
    for e in engs:
        e.set_training_loss(kind)
        e.set_target_shift(tshift)
        e.set_bn_shift(xshift)
    HybridEngine.comm_init_local(engs)
    ref = util.load_engine(spec, theta, X, f, y); ref.opt_init("Descent", 0.05)
    ref.set_training_loss(kind)
    per = B // world
    for s in range(3):
        a = (s % 2) * (per // 2)
        loss = HybridEngine.dp_train_step_group(engs, [a] * world, per // 2, want_loss=True)
        idx = np.concatenate([np.arange(r * per + a, r * per + a + per // 2) for r in range(world)]).astype(np.int32)
        lref = ref.train_step(0, idx.size, want_loss=True, idx=idx)
        assert abs(loss - lref) <= 1e-4 * abs(lref), (s, loss, lref)
    assert np.array_equal(engs[0].get_params(), engs[1].get_params())
    assert np.max(np.abs(engs[0].get_params() - ref.get_params())) <= 2e-5
    (m0, v0), (mr, vr) = engs[0].get_bn_state(), ref.get_bn_state()
    assert np.allclose(m0, mr, rtol=2e-6, atol=2e-6) and np.allclose(v0, vr, rtol=2e-5)
    for e in engs:
        e.close()
    ref.close()


@pytest.mark.parametrize("world", [2, 8])
def test_local_group_with_input_batchnorm_uses_the_statistics_of_the_global_minibatch(world):
    """input BatchNorm under the local group: the members' shifted sums are exchanged ahead of the pass (EH_BUF_BNSTAT), so every replica
    normalises with the mean / variance of the whole minibatch as one Lux BatchNorm would (src/models/NNModels.jl:97-105) and advances
    the same running statistics"""
    B = 2048
    spec, theta, X, f, y = util.rbq10_case(B, "tanh", True, 0.1)
    spec.input_batchnorm = True
    X = (X * np.float32(3.0) + np.float32(1.5)).astype(np.float32)
    engs = _shard_engines(spec, theta, X, f, y, world)
    shift = X.mean(axis=1).astype(np.float32)                      # the common shift of the sums (any vector, the same on every member)
    for e in engs:
        e.set_bn_shift(shift)
    HybridEngine.comm_init_local(engs)
    ref = util.load_engine(spec, theta, X, f, y); ref.opt_init("Adam", 0.01)
    per = B // world
    for s in range(4):
        a = (s % 2) * (per // 2)
        HybridEngine.dp_train_step_group(engs, [a] * world, per // 2)
        idx = np.concatenate([np.arange(r * per + a, r * per + a + per // 2) for r in range(world)]).astype(np.int32)
        ref.train_step(0, idx.size, want_loss=False, idx=idx)
    for e in engs[1:]:
        assert np.array_equal(engs[0].get_params(), e.get_params())
    assert np.max(np.abs(engs[0].get_params() - ref.get_params())) <= 3e-6
    (m0, v0), (mr, vr) = engs[0].get_bn_state(), ref.get_bn_state()
    assert np.allclose(m0, mr, rtol=2e-6, atol=2e-6) and np.allclose(v0, vr, rtol=2e-5)
    for e in engs:
        e.close()
    ref.close()


@pytest.mark.parametrize("world", [2, 8])
@pytest.mark.parametrize("kinds", ["pearsonLoss", "kgeLoss", ("pbkgeLoss", "mse"), ("rmse", "mae")])
def test_two_pass_losses_under_the_local_group(kinds, world):
    """pearson / kge / pbkge (and rmse on a multi-target model) under data parallelism (missing 3): the moments of the
    GLOBAL batch's predictions go round twice ahead of the pass (eh_dp_moments + EH_BUF_MOMENT), so the shards' gradient sums are
    those of ONE engine on the union -- with very different gaps per shard, where per-shard coefficients would be wrong.  Loss and
    gradient (through a Descent step) against the oracle on the whole batch; bar 1e-4 like every moment-based loss
    (src/losses/loss_fn.jl:105-174)."""
    rng = np.random.default_rng(8)
    B = 2048
    multi = not isinstance(kinds, str)
    if multi:
        pars = {"RUE": (0.1, 0.0, 1.0), "Rb": (1.0, 0.0, 6.0), "Q10": (1.5, 1.0, 4.0)}
        spec = ho.HybridSpec(6, [24, 12], "fluxpart", pars, ["RUE", "Rb"], ["Q10"], ["NEE", "GPP"], "tanh", True)
        X = rng.standard_normal((6, B)).astype(np.float32)
        f = {"SW_IN": (rng.random(B) * 400).astype(np.float32), "TA": (rng.random(B) * 30).astype(np.float32)}
        y = {"NEE": (2 + rng.standard_normal(B)).astype(np.float32), "GPP": (1 + rng.random(B) * 3).astype(np.float32)}
        y["NEE"][: B // 2][rng.random(B // 2) < 0.7] = np.nan
        y["GPP"][B // 2:][rng.random(B // 2) < 0.3] = np.nan
        theta = ho.init_theta(spec, 3, np.float32)
    else:
        spec, theta, X, f, y = util.rbq10_case(B, "tanh", True, 0.0, hidden=(24, 12))
        y["reco"][: B // world][::2] = np.nan
    engs = _shard_engines(spec, theta, X, f, y, world, opt=("Descent", 0.05))
    shift = [float(np.nanmean(y[t])) for t in spec.targets]
    for e in engs:
        e.set_training_loss(kinds if isinstance(kinds, str) else eh.PerTarget(kinds))
        e.set_target_shift(shift)
    HybridEngine.comm_init_local(engs)
    loss = HybridEngine.dp_train_step_group(engs, [0] * world, B // world, want_loss=True)
    l0, g0, _ = ho.loss_and_grad(spec, theta.astype(np.float64), X, f, y, kind=kinds if isinstance(kinds, str) else list(kinds))
    assert abs(loss - l0) <= 1e-4 * abs(l0), (loss, l0)
    step = (theta.astype(np.float64) - engs[0].get_params().astype(np.float64)) / 0.05
    assert util.relerr(step, g0) <= 1e-4, util.relerr(step, g0)
    for e in engs[1:]:
        assert np.array_equal(engs[0].get_params(), e.get_params())
    # and per-shard statistics would NOT have done: one engine alone on shard 0 sees a measurably different gradient direction
    with pytest.raises(eh.EngineError):
        engs[0].dp_grad(0, 64)                                  # the moments of this window have not gone round
    for e in engs:
        e.close()
