"""CPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/easyhybrid_hip.h declares, its structs match the ctypes mirror, and -- with no GPU in the
container -- every compute entry point fails loudly instead of falling back to a CPU path."""
import ctypes as C
import os
import re
import subprocess

import pytest

import easyhybrid_jl_amd as eh
from easyhybrid_jl_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "easyhybrid_hip.h")


def _declared():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(eh_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = L.lib()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
    assert sorted(L.SIGNATURES) == names, "ctypes SIGNATURES and the header disagree"
    assert lib.eh_version() == 4


def test_struct_layout_matches_header(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "easyhybrid_hip.h"\n'
                   'int main(){printf("%zu %zu %zu %zu\\n", sizeof(eh_model_desc), sizeof(eh_target_metrics),'
                   ' offsetof(eh_model_desc, param_default), offsetof(eh_model_desc, n_targets));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    a, b, c, d = map(int, subprocess.check_output([str(exe)]).split())
    assert a == C.sizeof(L.ModelDesc) and b == C.sizeof(L.TargetMetrics)
    assert c == L.ModelDesc.param_default.offset and d == L.ModelDesc.n_targets.offset


def test_julia_shim_struct_mirrors_the_ctypes_layout():
    """No `julia` in this image: the shim's `struct EhModelDesc` (what `@ccall eh_create` hands over) is checked field by field --
    name, order, element type, length -- against the ctypes mirror that the test above pins to the header."""
    jl = open(os.path.join(ROOT, "easyhybrid.jl_amd", "julia", "EasyHybridHIP", "src", "EasyHybridHIP.jl")).read()
    body = re.search(r"struct EhModelDesc\n(.*?)\nend", jl, flags=re.S).group(1)
    jt = {"Int32": (C.c_int32, 4), "UInt32": (C.c_uint32, 4), "Float32": (C.c_float, 4)}
    fields = []
    for line in body.splitlines():
        line = line.split("#")[0].strip()
        if not line:
            continue
        name, typ = [x.strip() for x in line.split("::")]
        m = re.fullmatch(r"NTuple\{(\d+), (\w+)\}", typ)
        fields.append((name, jt[m.group(2) if m else typ][0], int(m.group(1)) if m else 1))
    assert [f[0] for f in fields] == [n for n, _ in L.ModelDesc._fields_]
    off = 0
    for (name, ct, count), (_, pyt) in zip(fields, L.ModelDesc._fields_):
        assert getattr(L.ModelDesc, name).offset == off and C.sizeof(pyt) == 4 * count, name
        base = pyt
        while hasattr(base, "_type_") and not isinstance(base._type_, str):
            base = base._type_
        assert base is ct, name
        off += 4 * count
    assert off == C.sizeof(L.ModelDesc)


def _model(**kw):
    args = dict(hidden_layers=[16, 16])
    args.update(kw)
    return eh.constructHybridModel(["sw_pot", "dsw_pot"], ["ta"], ["reco"], eh.RbQ10, {"rb": (3, 0, 13), "Q10": (2, 1, 4)}, ["rb"], ["Q10"], **args)


def _has_gpu():
    import torch
    return torch.cuda.is_available()


@pytest.mark.skipif(_has_gpu(), reason="checks the no-GPU failure mode")
def test_no_gpu_fails_loudly_no_cpu_fallback():
    with pytest.raises(eh.EngineError, match="no HIP device"):
        _model().engine()


def test_create_rejects_bad_descriptors_before_touching_a_device():
    lib = L.lib()
    h = C.c_void_p()
    d = _model().to_desc()
    d.mech = 99
    assert lib.eh_create(C.byref(d), C.byref(h)) == L.EH_EUNSUPPORTED and b"unknown mechanistic model" in lib.eh_last_error(None)
    d = _model().to_desc()
    d.struct_size = 4
    assert lib.eh_create(C.byref(d), C.byref(h)) == L.EH_EINVAL
    d = _model().to_desc()
    d.hidden[0] = 512; d.input_batchnorm = 1; d.n_predictors = 40      # wider than the fused kernels, and an input BatchNorm over more predictors than the normalisation block holds
    assert lib.eh_create(C.byref(d), C.byref(h)) == L.EH_EUNSUPPORTED and b"no kernel for" in lib.eh_last_error(None)
    d = _model().to_desc()
    d.param_kind[0] = L.PAR_FIXED                         # no neural parameter left
    assert lib.eh_create(C.byref(d), C.byref(h)) == L.EH_EINVAL
    assert lib.eh_create(None, C.byref(h)) == L.EH_EINVAL
    assert lib.eh_destroy(None) == L.EH_OK
    # EH_ACT_PER_NET on a single-network descriptor: net_activation[l] is hidden layer l's (hidden_layers::Chain, NNModels.jl:145-219)
    d = _model().to_desc()
    d.activation = L.EH_ACT_PER_NET
    d.net_activation[d.n_hidden - 1] = 23
    assert lib.eh_create(C.byref(d), C.byref(h)) == L.EH_EUNSUPPORTED and b"activation id 23 for hidden layer" in lib.eh_last_error(None)
    ch = eh.constructHybridModel(["a", "b"], ["ta"], ["reco"], eh.RbQ10, {"rb": (3, 0, 13), "Q10": (2, 1, 4)}, ["rb"], ["Q10"],
                                 hidden_layers=eh.Chain(eh.Dense(16, 32, "relu"), eh.Dense(32, 8, "sigmoid")), activation="tanh")
    d = ch.to_desc()
    assert (d.activation, d.n_nets, d.n_hidden, list(d.hidden[:3]), list(d.net_activation[:3])) == (
        L.EH_ACT_PER_NET, 0, 3, [16, 32, 8], [L.ACTIVATIONS["tanh"], L.ACTIVATIONS["relu"], L.ACTIVATIONS["sigmoid"]])
    assert ch.layer_activations == ["tanh", "relu", "sigmoid"] and ch.n_theta == (2 * 16 + 16) + (16 * 32 + 32) + (32 * 8 + 8) + (8 + 1) + 1
    same = eh.constructHybridModel(["a", "b"], ["ta"], ["reco"], eh.RbQ10, {"rb": (3, 0, 13), "Q10": (2, 1, 4)}, ["rb"], ["Q10"],
                                   hidden_layers=eh.Chain(eh.Dense(16, 32, "tanh")), activation="tanh")
    assert same.layer_activations is None and same.to_desc().activation == L.ACTIVATIONS["tanh"]        # one activation after all: the plain descriptor
    mm = eh.constructHybridModel({"rb": ["a"], "Q10": ["b"]}, ["ta"], ["reco"], eh.RbQ10, {"rb": (3, 0, 13), "Q10": (2, 1, 4)}, [],
                                 hidden_layers={"rb": [16], "Q10": [8, 4]}, activation={"rb": "relu", "Q10": "tanh"})
    d = mm.to_desc()
    d.net_activation[1] = 17
    assert lib.eh_create(C.byref(d), C.byref(h)) == L.EH_EUNSUPPORTED and b"activation id 17" in lib.eh_last_error(None)
    d = mm.to_desc()
    d.net_depth[0] = 3                                    # deeper than n_hidden
    assert lib.eh_create(C.byref(d), C.byref(h)) == L.EH_EINVAL and b"net_depth" in lib.eh_last_error(None)
    d = mm.to_desc()
    d.net_depth[1] = 1                                    # nobody is n_hidden deep any more
    assert lib.eh_create(C.byref(d), C.byref(h)) == L.EH_EINVAL and b"no net is that deep" in lib.eh_last_error(None)
    assert lib.eh_mech_loss_vjp(None, 4, 4, None, None, None, None, None, None, None, None, None) == L.EH_EINVAL


def test_descriptor_from_model():
    m = _model(activation="swish", scale_nn_outputs=True, hidden_layers=[32, 24, 8])
    d = m.to_desc(3)
    assert (d.device, d.n_predictors, d.n_hidden, list(d.hidden)[:3]) == (3, 2, 3, [32, 24, 8])
    assert d.activation == L.ACTIVATIONS["swish"] and d.scale_nn_outputs == 1 and d.mech == 0 and d.n_params == 2
    assert list(d.param_kind)[:2] == [L.PAR_NEURAL, L.PAR_GLOBAL] and list(d.param_index)[:2] == [0, 0]
    assert (d.param_default[0], d.param_lower[0], d.param_upper[0]) == (3.0, 0.0, 13.0)
    assert (d.n_forcings, d.forcing_index[0], d.n_targets, d.target_output[0]) == (1, 0, 1, 0)
    assert m.n_theta == (2 * 32 + 32) + (32 * 24 + 24) + (24 * 8 + 8) + (8 * 1 + 1) + 1


# ---- every @ccall site of the Julia shim against the ctypes signatures (which the tests above pin to the header) -----------------
def _julia_ccalls():
    """[(symbol, [argument type strings], return type, line)] of every `@ccall LIB[].eh_*(...)::T` in EasyHybridHIP.jl"""
    jl = open(os.path.join(ROOT, "easyhybrid.jl_amd", "julia", "EasyHybridHIP", "src", "EasyHybridHIP.jl")).read()
    out = []
    for m in re.finditer(r"@ccall LIB\[\]\.(eh_[a-z0-9_]+)\(", jl):
        i, depth, args, cur = m.end(), 1, [], ""
        while depth:
            c = jl[i]
            if c in "([{":
                depth += 1
            elif c in ")]}":
                depth -= 1
                if depth == 0:
                    break
            if c == "," and depth == 1:
                args.append(cur); cur = ""
            else:
                cur += c
            i += 1
        if cur.strip():
            args.append(cur)
        ret = re.match(r"::([A-Za-z0-9_{}]+)", jl[i + 1:]).group(1)
        types = []
        for a in args:
            d, cut = 0, None
            for k, c in enumerate(a):                      # the annotation after the last top-level `::`
                if c in "([{":
                    d += 1
                elif c in ")]}":
                    d -= 1
                elif d == 0 and a[k:k + 2] == "::":
                    cut = k
            assert cut is not None, (m.group(1), a)
            types.append(a[cut + 2:].strip())
        out.append((m.group(1), types, ret, jl.count("\n", 0, m.start()) + 1))
    return out


_JL_SCALAR = {"Int32": C.c_int32, "Int64": C.c_int64, "UInt64": C.c_uint64, "Float32": C.c_float, "Float64": C.c_double, "UInt32": C.c_uint32}
_JL_STRUCT = {"EhModelDesc": L.ModelDesc, "EhTargetMetrics": L.TargetMetrics}


def _jl_matches(jt, ct):
    """does the Julia ccall annotation `jt` describe the C type ctypes calls `ct`?"""
    if jt in _JL_SCALAR:
        return ct is _JL_SCALAR[jt]
    if jt == "Cstring":
        return ct is C.c_char_p
    m = re.fullmatch(r"(?:Ptr|Ref)\{(.+)\}", jt)
    if not m:
        return False
    inner = m.group(1)
    if ct is C.c_void_p:                                   # void*: any pointer
        return True
    if ct is C.c_char_p:
        return inner in ("UInt8", "Cchar")
    if not hasattr(ct, "_type_") or isinstance(ct._type_, str):
        return False
    pointee = ct._type_                                    # POINTER(pointee)
    if inner in _JL_SCALAR:
        return pointee is _JL_SCALAR[inner]
    if inner in _JL_STRUCT:
        return pointee is _JL_STRUCT[inner]
    if re.fullmatch(r"Ptr\{.+\}", inner) or inner == "Ptr{Cvoid}":      # array of pointers
        return pointee is C.c_void_p or (hasattr(pointee, "_type_") and not isinstance(pointee._type_, str) and _jl_matches(inner, pointee))
    return False


def test_every_julia_ccall_matches_the_c_signature():
    calls = _julia_ccalls()
    assert len(calls) >= 40, len(calls)
    seen = set()
    for name, types, ret, line in calls:
        assert name in L.SIGNATURES, f"EasyHybridHIP.jl:{line}: @ccall of {name}, which include/easyhybrid_hip.h does not declare"
        res, argtypes = L.SIGNATURES[name]
        assert len(types) == len(argtypes), f"EasyHybridHIP.jl:{line}: {name} takes {len(argtypes)} arguments, the @ccall passes {len(types)}"
        for k, (jt, ct) in enumerate(zip(types, argtypes)):
            assert _jl_matches(jt, ct), f"EasyHybridHIP.jl:{line}: {name} argument {k}: `{jt}` against {ct}"
        assert (ret == "Cstring" and res is C.c_char_p) or (ret in _JL_SCALAR and _JL_SCALAR[ret] is res), f"EasyHybridHIP.jl:{line}: {name} returns {res}, @ccall says {ret}"
        seen.add(name)
    # the entry points a training run of the shim goes through are all bound
    for must in ("eh_create", "eh_destroy", "eh_set_data", "eh_set_params", "eh_get_params", "eh_opt_init", "eh_train_step", "eh_train_epoch",
                 "eh_eval", "eh_forward", "eh_loss_and_grad", "eh_mech_loss_vjp", "eh_comm_init", "eh_comm_init_local", "eh_dp_train_step",
                 "eh_dp_train_step_group", "eh_get_bn_state", "eh_set_bn_state"):
        assert must in seen, must
