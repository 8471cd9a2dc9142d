"""-m gpu: the RCCL communicator inside the library (eh_comm_* / eh_dp_allreduce / eh_dp_train_step), through the C ABI, with
world = 1 -- all a one-GPU box allows (RCCL refuses two ranks on one device); the multi-rank arithmetic of the seam is covered on
the CPU by tests/test_dp_gloo.py and by the virtual-shard tests of test_gpu_parity.py."""
import numpy as np
import pytest

import easyhybrid_jl_amd as eh
from easyhybrid_jl_amd import _lib as L
from easyhybrid_jl_amd.engine import HybridEngine
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
