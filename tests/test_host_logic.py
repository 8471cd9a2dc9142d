"""Host-side mirror of the reference interface: constructor checks, parameter helpers, data
preparation / splitting and the bookkeeping of `train` (no GPU needed)."""
import numpy as np
import pytest

import easyhybrid_jl_amd as eh
from easyhybrid_jl_amd import dp

PARAMS = {"rb": (3.0, 0.0, 13.0), "Q10": (2.0, 1.0, 4.0)}


def model(**kw):
    return eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, PARAMS, ["rb"], ["Q10"], **kw)


def test_constructor_mirrors_reference_struct():
    m = model()
    assert isinstance(m, eh.SingleNNHybridModel) and eh.HybridModel is eh.SingleNNHybridModel
    assert m.hidden_layers == [32, 32] and m.activation == "tanh" and not m.scale_nn_outputs and m.start_from_default
    assert m.neural_param_names == ["rb"] and m.global_param_names == ["Q10"] and m.fixed_param_names == []
    m2 = eh.constructHybridModel(["sw_pot"], ["ta"], ["reco"], "RbQ10", PARAMS, ["rb"], [])     # Q10 neither neural nor global -> fixed
    assert m2.fixed_param_names == ["Q10"]


def test_constructor_errors_match_reference():
    with pytest.raises(AssertionError, match="neural_param_names"):           # GenericHybridModel.jl:110 / test_generic_hybrid_model.jl:569-586
        eh.constructHybridModel(["a"], ["ta"], ["reco"], eh.RbQ10, PARAMS, ["nope"], ["Q10"])
    with pytest.raises(TypeError, match="dict"):                              # an unregistered callable is recorded as a device program (tests/test_program.py)
        eh.constructHybridModel(["a"], ["ta"], ["reco"], lambda **kw: None, PARAMS, ["rb"], ["Q10"])
    with pytest.raises(NotImplementedError, match="registry"):
        eh.constructHybridModel(["a"], ["ta"], ["reco"], "NoSuchModel", PARAMS, ["rb"], ["Q10"])
    assert model(input_batchnorm=True).to_desc().input_batchnorm == 1
    with pytest.raises(ValueError, match="forcing"):
        eh.constructHybridModel(["a"], ["temp"], ["reco"], eh.RbQ10, PARAMS, ["rb"], ["Q10"])


def test_scaling_helpers_known_answers():
    hp = eh.build_parameters({"a": (1.0, 0.0, 2.0), "b": (2.0, 1.0, 3.0)})
    assert eh.scale_single_param("a", np.float32([0.0]), hp)[0] == pytest.approx(1.0)      # test_generic_hybrid_model.jl:109-117
    assert eh.scale_single_param("b", np.float32([0.0]), hp)[0] == pytest.approx(2.0)
    assert eh.scale_single_param_minmax("a", hp) == pytest.approx(0.0)                     # :119-126
    assert eh.hard_sigmoid(np.array([-10.0, 0.0, 10.0])).tolist() == [0.0, 0.5, 1.0]       # :23-35
    assert eh.inv_hard_sigmoid(0.7) == pytest.approx(1.0)


def test_initialparameters_layout_and_defaults():
    m = model(hidden_layers=[16, 16])
    th = m.initialparameters(0)
    assert th.dtype == np.float32 and th.size == 338
    assert th[-1] == pytest.approx(np.log(0.5), rel=1e-6)          # start_from_default: inv_sigmoid((2-1)/(4-1))
    layers, glob = m.unpack(th)
    assert [w.shape for w, _ in layers] == [(16, 2), (16, 16), (1, 16)] and list(glob) == ["Q10"]


def test_prepare_data_drops_rows_like_reference():
    n = 10
    cols = {"sw_pot": np.arange(n, dtype=float), "dsw_pot": np.ones(n), "ta": np.linspace(0, 9, n), "reco": np.arange(n, dtype=float), "id": np.arange(n)}
    cols["sw_pot"][2] = np.nan            # predictor missing -> row dropped (prepare_data.jl:45-55)
    cols["reco"][5] = np.nan              # only target missing and it is the only target -> dropped
    (X, f), y = eh.prepare_data(model(), cols)
    assert X.shape == (2, 8) and X.dtype == np.float32 and f["ta"].shape == (8,) and not np.isnan(y["reco"]).any()


def test_split_data_ratio_and_folds():
    n = 100
    cols = {"sw_pot": np.arange(n, dtype=float), "dsw_pot": np.ones(n), "ta": np.zeros(n), "reco": np.ones(n)}
    (tr, _), (va, _) = [(a[0][0], a[1]) for a in eh.split_data(cols, model())]
    assert tr.shape[1] == 80 and va.shape[1] == 20 and tr[0, 0] == 0 and va[0, 0] == 80        # chronological, at = 0.8
    folds = np.arange(n) % 5 + 1
    (tr, _), (va, _) = [(a[0][0], a[1]) for a in eh.split_data(cols, model(), eh.DataConfig(folds=folds, val_fold=2))]
    assert va.shape[1] == 20 and tr.shape[1] == 80 and set(va[0].astype(int) % 5) == {1}
    with pytest.raises(ValueError):
        eh.split_data(cols, model(), eh.DataConfig(folds=folds, val_fold=1, split_by_id=folds))


def test_config_validation_and_directions():
    assert eh.isbetter(0.1, 0.2, "mse") and eh.isbetter(0.9, 0.8, "r2") and not eh.isbetter(0.9, 0.8, "rmse")   # loss_fn.jl:181-194
    with pytest.raises(ValueError, match="to be maximized"):
        eh.check_training_loss("nse")                                                                         # loss_fn.jl:196-205
    eh.validate_config(eh.TrainConfig())
    with pytest.raises(ValueError):
        eh.validate_config(eh.TrainConfig(batchsize=0))
    eh.validate_config(eh.TrainConfig(training_loss="nseLoss"))
    eh.validate_config(eh.TrainConfig(training_loss="kgeLoss"))                 # two-pass losses are built
    eh.validate_config(eh.TrainConfig(training_loss=lambda a, b: np.mean(np.abs(a - b))))      # a function is recorded when the engine is set up (tests/test_program.py)
    with pytest.raises(NotImplementedError):
        eh.validate_config(eh.TrainConfig(training_loss="no_such_loss"))
    with pytest.raises(TypeError):
        eh.train(model(), {}, bogus_keyword=1)


def test_shard_ranges_cover_exactly():
    for n, w in [(10, 3), (65536 * 8, 8), (7, 8)]:
        r = [dp.shard_range(n, k, w) for k in range(w)]
        assert r[0][0] == 0 and r[-1][1] == n and all(r[i][1] == r[i + 1][0] for i in range(w - 1))
        assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_multinn_constructor_mirrors_reference():
    # constructHybridModel(predictors::NamedTuple, forcing, targets, f, parameters, global_param_names; ...)  GenericHybridModel.jl:142-206
    m = eh.constructHybridModel({"rb": ["sw_pot", "dsw_pot"], "Q10": ["dsw_pot"]}, ["ta"], ["reco"], eh.RbQ10, PARAMS, [],
                                hidden_layers={"rb": [16, 16], "Q10": [8, 4]}, activation="tanh")
    assert m.neural_param_names == ["rb", "Q10"] and m.global_param_names == [] and m.predictors == ["sw_pot", "dsw_pot", "dsw_pot"]
    assert m.NNs["rb"] == [(16, 2), (16, 16), (1, 16)] and m.NNs["Q10"] == [(8, 1), (4, 8), (1, 4)]
    assert m.n_theta == (32 + 16 + 256 + 16 + 16 + 1) + (8 + 8 + 32 + 4 + 4 + 1)
    d = m.to_desc()
    assert d.n_nets == 2 and list(d.net_n_predictors)[:2] == [2, 1] and list(d.net_hidden[1])[:2] == [8, 4] and d.n_predictors == 3
    th = m.initialparameters(0)
    nets, glob = m.unpack(th)
    assert th.size == m.n_theta and nets["Q10"][0][0].shape == (8, 1) and glob == {}
    # nets of different depth (test/test_generic_hybrid_model.jl:346): theta holds each net with its own layers; the envelope
    # carries the shallower one through an identity block as wide as its last hidden layer
    m2 = eh.constructHybridModel({"rb": ["a"], "Q10": ["b"]}, ["ta"], ["reco"], eh.RbQ10, PARAMS, [], hidden_layers={"rb": [16], "Q10": [8, 4]})
    assert m2.NNs["rb"] == [(16, 1), (1, 16)] and m2.NNs["Q10"] == [(8, 1), (4, 8), (1, 4)] and m2.hidden_layers == [24, 20]
    assert m2.n_theta == (16 + 16 + 16 + 1) + (8 + 8 + 32 + 4 + 4 + 1)
    d2 = m2.to_desc()
    assert d2.n_hidden == 2 and list(d2.net_depth)[:2] == [1, 2] and list(d2.net_hidden[0])[:2] == [16, 0]


def test_multinn_per_network_activations():
    # activation::NamedTuple next to hidden_layers::NamedTuple (GenericHybridModel.jl:168-176): net k gets activation[k]
    m = eh.constructHybridModel({"rb": ["sw_pot", "dsw_pot"], "Q10": ["dsw_pot"]}, ["ta"], ["reco"], eh.RbQ10, PARAMS, [],
                                hidden_layers={"rb": [16, 16], "Q10": [8, 4]}, activation={"rb": "swish", "Q10": "tanh"})
    assert m.net_activations == ["swish", "tanh"] and m.activation == {"rb": "swish", "Q10": "tanh"}
    d = m.to_desc()
    assert d.activation == eh._lib.EH_ACT_PER_NET and list(d.net_activation)[:2] == [eh._lib.ACTIVATIONS["swish"], eh._lib.ACTIVATIONS["tanh"]]
    assert m.initialparameters(0).size == m.n_theta
    # the same activation everywhere is the plain single-activation model (kernels built ahead of time)
    m1 = eh.constructHybridModel({"rb": ["a"], "Q10": ["b"]}, ["ta"], ["reco"], eh.RbQ10, PARAMS, [],
                                 hidden_layers={"rb": [16], "Q10": [8]}, activation={"rb": "relu", "Q10": "relu"})
    assert m1.net_activations is None and m1.activation == "relu" and m1.to_desc().activation == eh._lib.ACTIVATIONS["relu"]
    with pytest.raises(TypeError):         # the reference reads activation[nn_name] only when hidden_layers is a NamedTuple too
        eh.constructHybridModel({"rb": ["a"], "Q10": ["b"]}, ["ta"], ["reco"], eh.RbQ10, PARAMS, [], hidden_layers=[8, 8],
                                activation={"rb": "relu", "Q10": "tanh"})
    with pytest.raises(NotImplementedError):
        eh.constructHybridModel({"rb": ["a"], "Q10": ["b"]}, ["ta"], ["reco"], eh.RbQ10, PARAMS, [],
                                hidden_layers={"rb": [16], "Q10": [8]}, activation={"rb": "relu", "Q10": "gelu"})


def test_weight_l2_terms_fold_into_one_coefficient_per_entry():
    """extra_loss = (yhat, ps) -> (; l2_rb = a * weight_l2(ps.rb; normalize = true), l2_all = b * weight_l2(ps), bias = c * weight_l2(ps; key = :bias))
    (src/utils/extract_weights.jl:64-91): what the mirror hands to eh_set_weight_l2_coef against the oracle's term-by-term walk"""
    from oracle import hybrid_oracle as ho
    from tests import util
    from easyhybrid_jl_amd.train import _extra_terms, _extra_loss_values
    spec = ho.HybridSpec(4, [1], "rbq10", dict(ho.RBQ10_PARAMS), ["rb", "Q10"], [], ["reco"], "tanh", True, nets=[([0, 1], [8, 8]), ([2, 3], [16, 8])])
    model = util.model_from_spec(spec)
    theta = ho.init_theta(spec, 6, np.float64)
    terms = _extra_terms({"l2_rb": eh.WeightL2(0.3, normalize=True, net="rb"), "l2_all": eh.WeightL2(0.02), "bias": eh.WeightL2(0.1, key="bias", net="Q10")})
    coef = model.l2_coefficients(terms).astype(np.float64)
    vals, grad = ho.weight_l2_terms(spec, theta, [(0.3, True, 0, "weight"), (0.02, False, None, "weight"), (0.1, False, 1, "bias")])
    assert np.sum(coef * theta * theta) == pytest.approx(sum(vals), rel=1e-6)
    assert np.allclose(2 * coef * theta, grad, rtol=1e-6, atol=1e-12)
    hv = _extra_loss_values(model, theta, terms)
    assert [hv["l2_rb"], hv["l2_all"], hv["bias"]] == pytest.approx(vals, rel=1e-12) and hv["sum"] == pytest.approx(sum(vals), rel=1e-12)
    # the single whole-tree term is weight_mask's
    assert np.array_equal(model.l2_mask(), model.weight_mask()) and np.array_equal(model.l2_mask(), ho.weight_mask(spec))
    assert not np.any(model.l2_mask(key="bias") & model.weight_mask())
    with pytest.raises(KeyError):
        model.l2_mask(net="nope")
    # a function of the predictions is no WeightL2 term: it is recorded when the engine is created (train._extra_fn / program.trace_extra_loss)
    from easyhybrid_jl_amd.train import _extra_fn
    fn = lambda yhat, ps: [np.sum(np.abs(yhat["reco"]))]
    assert _extra_terms(fn) == [] and _extra_fn(fn) is fn and _extra_fn([eh.WeightL2(0.1), fn]) is fn and len(_extra_terms([eh.WeightL2(0.1), fn])) == 1
    with pytest.raises(NotImplementedError, match="extra_loss"):
        _extra_terms("l2")
    from easyhybrid_jl_amd.program import trace_extra_loss
    ent = trace_extra_loss(lambda yh, ps: {"a": np.sum(np.abs(yh["v1"])), "b": 0.5 * np.mean(yh["v2"] ** 2)}, ["v1", "v2"])
    assert [(e[0], e[1], e[2]) for e in ent] == [("a", "v1", "sum"), ("b", "v2", "mean")] and ent[0][3].words() == [13]
    with pytest.raises(NotImplementedError, match="ONE output"):
        trace_extra_loss(lambda yh: [np.sum(yh["v1"] * yh["v2"])], ["v1", "v2"])
    with pytest.raises(NotImplementedError):
        trace_extra_loss(lambda yh: [np.sum(yh["v1"]) + 1.0], ["v1", "v2"])          # a constant added to a sum: once per sample? refused
    with pytest.raises(NotImplementedError, match="WeightL2"):
        trace_extra_loss(lambda yh, ps: [np.sum(yh["v1"]) * ps.w], ["v1"])
    with pytest.raises(ValueError, match="same name"):
        _extra_terms([eh.WeightL2(0.1), eh.WeightL2(0.2)])


def test_hidden_layers_given_as_a_chain_of_dense_layers():
    """hidden_layers::Chain (NNModels.jl:145-219): the reference wraps the user's hidden layers as Dense(in, first_h, act) -> layers ->
    Dense(last_h, out); a chain of Dense layers on the model's activation is the vector [first_h, out_1, ..., out_n]"""
    from easyhybrid_jl_amd.synthetic import RBQ10_PARAMS
    kw = dict(activation="tanh", scale_nn_outputs=True)
    args = (["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, dict(RBQ10_PARAMS), ["rb"], ["Q10"])
    m_chain = eh.constructHybridModel(*args, hidden_layers=eh.Chain(eh.Dense(16, 16, "tanh"), eh.Dense(16, 8, np.tanh)), **kw)
    m_vec = eh.constructHybridModel(*args, hidden_layers=[16, 16, 8], **kw)
    assert m_chain.NN == m_vec.NN == [(16, 2), (16, 16), (8, 16), (1, 8)]
    assert m_chain.n_theta == m_vec.n_theta
    np.testing.assert_array_equal(m_chain.initialparameters(3), m_vec.initialparameters(3))
    # per network of a MultiNNHybridModel (GenericHybridModel.jl:168-176 hands hidden_layers[nn_name] to the same function)
    mm = eh.constructHybridModel({"rb": ["sw_pot", "dsw_pot"]}, ["ta"], ["reco"], eh.RbQ10, dict(RBQ10_PARAMS), ["Q10"],
                                 hidden_layers={"rb": eh.Chain(eh.Dense(8, 4, "sigmoid"))}, activation={"rb": "sigmoid"})
    assert mm.config["hidden_layers"] == {"rb": [8, 4]}
    # a single network takes an activation per layer: [model's, layer 1's, ...]; the gains of the initialisation follow the layers
    m_mixed = eh.constructHybridModel(*args, hidden_layers=eh.Chain(eh.Dense(16, 16, "relu")), **kw)
    assert m_mixed.layer_activations == ["tanh", "relu"] and m_mixed.NN == [(16, 2), (16, 16), (1, 16)]
    assert m_mixed.activation_of(0, 1) == "relu" and m_mixed.config["layer_activations"] == ["tanh", "relu"]
    # ... the networks of a MultiNN model one per NETWORK
    with pytest.raises(NotImplementedError, match="ONE activation"):
        eh.constructHybridModel({"rb": ["sw_pot", "dsw_pot"]}, ["ta"], ["reco"], eh.RbQ10, dict(RBQ10_PARAMS), ["Q10"],
                                hidden_layers={"rb": eh.Chain(eh.Dense(8, 4, "relu"))}, activation={"rb": "sigmoid"})
    with pytest.raises(ValueError, match="takes 8 inputs"):
        eh.constructHybridModel(*args, hidden_layers=eh.Chain(eh.Dense(16, 16, "tanh"), eh.Dense(8, 8, "tanh")), **kw)
    with pytest.raises(NotImplementedError, match="only Dense"):
        eh.constructHybridModel(*args, hidden_layers=eh.Chain(object()), **kw)
    with pytest.raises(ValueError, match="empty Chain"):
        eh.constructHybridModel(*args, hidden_layers=eh.Chain(), **kw)
