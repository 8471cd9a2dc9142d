# EasyHybridHIP.jl -- thin Julia host over libeasyhybrid_hip.so (include/easyhybrid_hip.h).
#
# Keeps the reference's front door -- `constructHybridModel(...)` and `train(model, data; ...)`
# (EasyHybrid.jl src/models/GenericHybridModel.jl:89-140, src/training/train.jl:211-219) -- and
# replaces what `run_epoch!` / `evaluate_epoch` do per minibatch / per epoch
# (src/training/epoch.jl:13-33, 53-66) by `@ccall`s into the HIP engine.  No CUDA.jl, no AMDGPU.jl:
# the only thing that crosses the boundary is plain pointers and sizes.
#
# NOTE: the build container has no `julia`, so this file is written against the C header and
# mirrored call-for-call by the ctypes harness (easyhybrid.jl_amd/_lib.py, engine.py), which IS
# exercised by the test-suite.  INTEGRATION.md shows the same binding inside EasyHybrid itself.
module EasyHybridHIP

using Libdl
using Random

export PerTarget, set_training_loss!, set_agg!, set_weight_l2!, set_weight_l2_coef!, constructHybridModel, SingleNNHybridModel, MultiNNHybridModel, HybridModel, train, train!, HybridEngine, prepare_data, split_data, initialparameters,
    Adam, AdamW, RMSProp, Descent, RbQ10, Expo_resp_model,
    LinearHM, Expo2Pool, Rs_components, Rs_components3F, FluxPartModelQ10

const LIB = Ref{String}(get(ENV, "EASYHYBRID_HIP_LIB", joinpath(@__DIR__, "..", "..", "..", "libeasyhybrid_hip.so")))

const EH_MAX_HIDDEN, EH_MAX_PARAMS, EH_MAX_FORC, EH_MAX_TARG = 8, 8, 4, 4
const EH_SPLIT_TRAIN, EH_SPLIT_VAL = Int32(0), Int32(1)

# mirror of `eh_model_desc` (field order and widths exactly as in the header)
struct EhModelDesc
    struct_size::Int32
    device::Int32
    n_predictors::Int32
    n_hidden::Int32
    hidden::NTuple{8, Int32}
    activation::Int32
    scale_nn_outputs::Int32
    input_batchnorm::Int32
    mech::Int32
    n_params::Int32
    param_kind::NTuple{8, Int32}
    param_index::NTuple{8, Int32}
    param_default::NTuple{8, Float32}
    param_lower::NTuple{8, Float32}
    param_upper::NTuple{8, Float32}
    n_forcings::Int32
    forcing_index::NTuple{4, Int32}
    n_targets::Int32
    target_output::NTuple{4, Int32}
    n_nets::Int32                                   # 0 = SingleNNHybridModel
    net_n_predictors::NTuple{8, Int32}
    net_hidden::NTuple{64, Int32}                   # [8 nets][8 layers], row-major like the C array
    net_activation::NTuple{8, Int32}                # read when activation == 5 (EH_ACT_PER_NET): activation id of net k
    net_depth::NTuple{8, Int32}                     # hidden layers of net k (0 = n_hidden); n_hidden is the deepest net's
    prog_len::Int32                                 # EH_MECH_PROGRAM only (a recorded closure, see `record_program`)
    prog_n_const::Int32
    prog_n_forc::Int32
    prog_n_out::Int32
    prog_out::NTuple{3, Int32}
    prog_code::NTuple{64, UInt32}
    prog_const::NTuple{16, Float32}
end

struct EhTargetMetrics
    n::Float64; mse::Float64; rmse::Float64; mae::Float64; r2::Float64; nse::Float64
    pearson::Float64; kge::Float64; pbkge::Float64; beta::Float64; alpha::Float64; sse::Float64
end

# ------------------------------------------------------------------------------------------------
# mechanistic registry: the Julia functions of the reference, tagged with the device model id
# ------------------------------------------------------------------------------------------------
struct Program                       # include/easyhybrid_hip.h `eh_prog_op`: slots 0..7 params, 8..11 forcings, 12..27 constants, 28+i instruction i
    consts::Vector{Float32}
    code::Vector{UInt32}             # op | a << 8 | b << 16 | c << 24
    out::Vector{Int32}
end
struct MechSpec
    id::Int32
    params::Vector{Symbol}
    forcings::Vector{Symbol}
    outputs::Vector{Symbol}
    program::Union{Nothing, Program}
end
MechSpec(id, params, forcings, outputs) = MechSpec(id, params, forcings, outputs, nothing)

RbQ10(; ta, Q10, rb, tref = 15.0f0) = (; reco = rb .* Q10 .^ (0.1f0 .* (ta .- tref)), Q10, rb)       # test/test_split_data_train.jl:36-39
Expo_resp_model(; T, Resp0, k) = (; Resp_obs = Resp0 .* exp.(k .* T), Resp0, k)                      # projects/ExpoHybrid/ExpoHybridEstim.jl:69-85
LinearHM(; x, alpha, beta) = (; obs = alpha .* x .+ beta, alpha, beta)                                   # src/models/LinearHM.jl:61-68
Expo2Pool(; T, R0a, ka, R0b, kb) = (; Resp_obs = R0a .* exp.(ka .* T) .+ R0b .* exp.(kb .* T), R0a, ka, R0b, kb)   # build-defined (BASELINE config 3)
function Rs_components(; ta, Rb_het, Rb_root, Rb_myc, Q10_het, Q10_root, Q10_myc, tref = 15.0f0)       # src/models/Rs_components.jl:40-57
    e = 0.1f0 .* (ta .- tref)
    R_het = Rb_het .* Q10_het .^ e; R_root = Rb_root .* Q10_root .^ e; R_myc = Rb_myc .* Q10_myc .^ e
    return (; R_soil = R_het .+ R_root .+ R_myc, R_het, R_root, R_myc)
end
function Rs_components3F(; ta, sw_in, vpd, Rb_het, Rb_root, Rb_myc, Q10_het, Q10_root, Q10_myc, tref = 15.0f0)   # build-defined (BASELINE config 5)
    e = 0.1f0 .* (ta .- tref)
    R_het = Rb_het .* Q10_het .^ e; R_root = sw_in .* Rb_root .* Q10_root .^ e; R_myc = vpd .* Rb_myc .* Q10_myc .^ e
    return (; R_soil = R_het .+ R_root .+ R_myc, R_het, R_root, R_myc)
end
function FluxPartModelQ10(; SW_IN, TA, RUE, Rb, Q10, tref = 15.0f0)                                     # src/models/FluxPartModel_Q10_Lux.jl:66-74
    GPP = SW_IN .* RUE ./ 12.011f0
    RECO = Rb .* Q10 .^ (0.1f0 .* (TA .- tref))
    return (; NEE = RECO .- GPP, GPP, RECO)
end
const MECH = IdDict{Any, MechSpec}(
    RbQ10 => MechSpec(0, [:rb, :Q10], [:ta], [:reco]),
    Expo_resp_model => MechSpec(1, [:Resp0, :k], [:T], [:Resp_obs]),
    LinearHM => MechSpec(2, [:alpha, :beta], [:x], [:obs]),
    Expo2Pool => MechSpec(3, [:R0a, :ka, :R0b, :kb], [:T], [:Resp_obs]),
    Rs_components => MechSpec(4, [:Rb_het, :Rb_root, :Rb_myc, :Q10_het, :Q10_root, :Q10_myc], [:ta], [:R_soil]),
    Rs_components3F => MechSpec(7, [:Rb_het, :Rb_root, :Rb_myc, :Q10_het, :Q10_root, :Q10_myc], [:ta, :sw_in, :vpd], [:R_soil]),
    FluxPartModelQ10 => MechSpec(5, [:RUE, :Rb, :Q10], [:SW_IN, :TA], [:NEE, :GPP, :RECO]),
)
"Register another Julia function under one of the device model ids (see include/easyhybrid_hip.h)."
register_mechanistic!(f, spec::MechSpec) = (MECH[f] = spec)

# ------------------------------------------------------------------------------------------------
# any other closure `f(; forcing..., params...) -> NamedTuple` (GenericHybridModel.jl:420-425): called ONCE with tracer
# numbers, which record its elementwise arithmetic as a straight-line program; the step kernel evaluates the program per
# sample and runs the reverse sweep over the same tape (what Zygote derives from the closure).  Broadcast dots work as
# they are (a tracer is a scalar).  Cross-sample operations (sum, mean, cumsum) and `x > 0 ? a : b` cannot be recorded --
# use `ifelse.(x .> 0, a, b)`.
# ------------------------------------------------------------------------------------------------
const OPS = Dict(:add => 0, :sub => 1, :mul => 2, :div => 3, :neg => 4, :exp => 5, :log => 6, :pow => 7, :sqrt => 8, :tanh => 9,
                 :sigmoid => 10, :max => 11, :min => 12, :abs => 13, :sin => 14, :cos => 15, :select => 16, :gt => 17)
struct Tape
    nodes::Vector{Tuple}             # (:par, j) | (:frc, name) | (:const, value) | (op, a, b, c) with node ids
    index::Dict{Tuple, Int}
end
Tape() = Tape(Tuple[], Dict{Tuple, Int}())
node!(t::Tape, k::Tuple) = get!(() -> (push!(t.nodes, k); length(t.nodes)), t.index, k)
struct Tr <: Real                    # a traced per-sample value
    tape::Tape
    id::Int
end
struct TrBool                        # a traced comparison: only `ifelse` takes it
    v::Tr
end
lift(t::Tape, x::Tr) = x
lift(t::Tape, x::Real) = (isfinite(x) || throw(ArgumentError("constant $x in a mechanistic program")); Tr(t, node!(t, (:const, Float32(x)))))
rec(op::Symbol, a::Tr, rest...) = Tr(a.tape, node!(a.tape, (op, a.id, (lift(a.tape, r).id for r in rest)...)))
for (f, op) in ((:+, :add), (:-, :sub), (:*, :mul), (:/, :div), (:max, :max), (:min, :min))
    @eval Base.$f(a::Tr, b::Tr) = rec($(QuoteNode(op)), a, b)
    @eval Base.$f(a::Tr, b::Real) = rec($(QuoteNode(op)), a, b)
    @eval Base.$f(a::Real, b::Tr) = rec($(QuoteNode(op)), lift(b.tape, a), b)
end
for (f, op) in ((:-, :neg), (:exp, :exp), (:log, :log), (:sqrt, :sqrt), (:tanh, :tanh), (:abs, :abs), (:sin, :sin), (:cos, :cos))
    @eval Base.$f(a::Tr) = rec($(QuoteNode(op)), a)
end
Base.:+(a::Tr) = a
Base.inv(a::Tr) = 1.0f0 / a
Base.abs2(a::Tr) = a * a
Base.exp2(a::Tr) = 2.0f0^a
Base.log2(a::Tr) = log(a) * Float32(1 / log(2))
Base.log10(a::Tr) = log(a) * Float32(1 / log(10))
sigmoid(a::Tr) = rec(:sigmoid, a)
Base.:^(a::Tr, b::Tr) = rec(:pow, a, b)                                   # base > 0
function Base.:^(a::Tr, n::Integer)                                       # products: valid for negative bases too
    n == 0 && return lift(a.tape, 1.0f0)
    r = nothing; b = a; k = abs(n)
    while k > 0
        isodd(k) && (r = r === nothing ? b : r * b)
        k >>= 1
        k > 0 && (b = b * b)
    end
    return n > 0 ? r : inv(r)
end
Base.:^(a::Tr, b::Real) = isinteger(b) && abs(b) <= 8 ? a^Int(b) : (b == 0.5 ? sqrt(a) : rec(:pow, a, b))
Base.:^(a::Real, b::Tr) = (a > 0 || throw(ArgumentError("power with the non-positive constant base $a")); rec(:pow, lift(b.tape, a), b))
gt(a::Tr, b::Tr) = TrBool(rec(:gt, a, b))
gt(a::Tr, b::Real) = TrBool(rec(:gt, a, b))
gt(a::Real, b::Tr) = TrBool(rec(:gt, lift(b.tape, a), b))
ge(a, b) = TrBool(1.0f0 - gt(b, a).v)                                     # a >= b  <=>  !(b > a)
for (A, B) in ((:Tr, :Tr), (:Tr, :Real), (:Real, :Tr))                    # (every pair spelled out: Tr <: Real, so a Union would be ambiguous)
    @eval Base.:>(a::$A, b::$B) = gt(a, b)
    @eval Base.:<(a::$A, b::$B) = gt(b, a)
    @eval Base.:>=(a::$A, b::$B) = ge(a, b)
    @eval Base.:<=(a::$A, b::$B) = ge(b, a)
end
Base.ifelse(c::TrBool, a::Union{Tr, Real}, b::Union{Tr, Real}) = rec(:select, c.v, a, b)
Base.clamp(x::Tr, lo::Real, hi::Real) = min(max(x, lo), hi)

"Record `f(; forcing..., params...)` and encode what the `targets` outputs depend on."
function record_program(f, params::Vector{Symbol}, forcing::Vector{Symbol}, targets::Vector{Symbol})
    1 <= length(params) <= EH_MAX_PARAMS || throw(ArgumentError("a mechanistic program takes 1..$EH_MAX_PARAMS parameters"))
    t = Tape()
    kw = merge(NamedTuple{Tuple(forcing)}(Tuple(Tr(t, node!(t, (:frc, n))) for n in forcing)),
               NamedTuple{Tuple(params)}(Tuple(Tr(t, node!(t, (:par, j - 1))) for j in eachindex(params))))
    res = f(; kw...)
    outs = unique(targets)
    all(o -> haskey(res, o), outs) || throw(ArgumentError("targets $targets are not all outputs of the mechanistic model $(keys(res))"))
    length(outs) <= 3 || throw(ArgumentError("$(length(outs)) distinct target outputs (device limit 3)"))
    used = Symbol[]; consts = Float32[]; code = UInt32[]; slot = Dict{Int, Int}()
    function emit(id::Int)
        haskey(slot, id) && return slot[id]
        n = t.nodes[id]
        s = if n[1] === :par
            n[2]
        elseif n[1] === :frc
            n[2] in used || push!(used, n[2]); 8 + findfirst(==(n[2]), used) - 1
        elseif n[1] === :const
            n[2] in consts || push!(consts, n[2]); 12 + findfirst(==(n[2]), consts) - 1
        else
            a = [emit(i) for i in n[2:end]]; append!(a, zeros(Int, 3 - length(a)))
            push!(code, UInt32(OPS[n[1]]) | UInt32(a[1]) << 8 | UInt32(a[2]) << 16 | UInt32(a[3]) << 24)
            28 + length(code) - 1
        end
        return slot[id] = s
    end
    out = Int32[]
    for o in outs
        v = res[o]
        v isa Tr || (v = lift(t, v))
        s = emit(v.id)
        if s < 28                         # an output that is a bare input: x + 0 gives it an instruction
            z = emit(lift(t, 0.0f0).id)
            push!(code, UInt32(OPS[:add]) | UInt32(s) << 8 | UInt32(z) << 16); s = 28 + length(code) - 1
        end
        push!(out, s)
    end
    length(used) <= EH_MAX_FORC && length(consts) <= 16 && length(code) <= 64 ||
        throw(ArgumentError("program of $(length(code)) operations / $(length(consts)) constants / $(length(used)) forcings exceeds the device limits 64 / 16 / $EH_MAX_FORC"))
    return MechSpec(6, params, used, outs, Program(consts, code, out))
end

const ACT = Dict(:tanh => 0, :sigmoid => 1, :relu => 2, :swish => 3, :identity => 4)

# ------------------------------------------------------------------------------------------------
# model (field-for-field the reference's SingleNNHybridModel, GenericHybridModel.jl:44-63, minus Lux)
# ------------------------------------------------------------------------------------------------
struct SingleNNHybridModel
    NN::Vector{Tuple{Int, Int}}          # Dense (out, in) shapes that prepare_hidden_chain would build
    predictors::Vector{Symbol}
    forcing::Vector{Symbol}
    targets::Vector{Symbol}
    mechanistic_model::Function
    parameters::NamedTuple               # name => (default, lower, upper)
    neural_param_names::Vector{Symbol}
    global_param_names::Vector{Symbol}
    fixed_param_names::Vector{Symbol}
    scale_nn_outputs::Bool
    start_from_default::Bool
    config::NamedTuple
end
const HybridModel = SingleNNHybridModel

function constructHybridModel(predictors::Vector{Symbol}, forcing, targets, mechanistic_model, parameters,
        neural_param_names, global_param_names; hidden_layers::Vector{Int} = [32, 32], activation = tanh,
        scale_nn_outputs = false, input_batchnorm = false, start_from_default = true, kwargs...)
    all_names = collect(keys(parameters))
    # a closure outside the registry is recorded as a device program (no CPU fallback either way)
    haskey(MECH, mechanistic_model) || (MECH[mechanistic_model] = record_program(mechanistic_model, all_names, collect(Symbol, forcing), collect(Symbol, targets)))
    @assert all(n in all_names for n in neural_param_names) "neural_param_names ⊆ param_names"
    dims = [length(predictors); hidden_layers; length(neural_param_names)]
    NN = [(dims[i + 1], dims[i]) for i in 1:(length(dims) - 1)]
    fixed = [n for n in all_names if !(n in [neural_param_names..., global_param_names...])]
    config = (; hidden_layers, activation = Symbol(nameof(activation)), scale_nn_outputs, input_batchnorm, start_from_default, kwargs...)
    return SingleNNHybridModel(NN, predictors, collect(forcing), collect(targets), mechanistic_model, parameters,
        collect(neural_param_names), collect(global_param_names), fixed, scale_nn_outputs, start_from_default, config)
end

"""
    constructHybridModel(predictors::NamedTuple, forcing, targets, mechanistic_model, parameters, global_param_names; ...)

MultiNNHybridModel form (GenericHybridModel.jl:142-206): `predictors = (rb = [:sw_pot, :dsw_pot], Q10 = [:ta_lag], ...)` gives every
neural parameter its own single-output MLP on its own predictor columns; `hidden_layers` and `activation` may be NamedTuples
over the same keys.  The device runs the nets as ONE block-diagonal MLP (`NN` below is that envelope); `predictors` of the
returned model is the per-net predictor lists one after the other, and `X` handed to `set_data!` has its rows in that order
(a column used by two nets appears twice).  Different activations per net (`activation = 5`, `net_activation[k]` in the
descriptor) or different depths (`net_depth[k]`) select kernels compiled at run time.
"""
function constructHybridModel(predictors::NamedTuple, forcing, targets, mechanistic_model, parameters, global_param_names;
        hidden_layers::Union{Vector{Int}, NamedTuple} = [32, 32], activation::Union{Function, NamedTuple} = tanh,
        scale_nn_outputs = false, input_batchnorm = false, start_from_default = true, kwargs...)
    all_names = collect(keys(parameters))
    neural = collect(Symbol, keys(predictors))
    haskey(MECH, mechanistic_model) || (MECH[mechanistic_model] = record_program(mechanistic_model, all_names, collect(Symbol, forcing), collect(Symbol, targets)))
    @assert all(n in all_names for n in neural) "neural_param_names ⊆ param_names"
    # the reference reads activation[nn_name] only next to hidden_layers[nn_name] (GenericHybridModel.jl:168-176)
    activation isa NamedTuple && !(hidden_layers isa NamedTuple) &&
        throw(ArgumentError("activation given per network needs hidden_layers given per network as well"))
    hl = [hidden_layers isa NamedTuple ? collect(Int, hidden_layers[k]) : collect(Int, hidden_layers) for k in neural]
    all(!isempty, hl) || throw(ArgumentError("unsupported: a network without a hidden layer"))
    acts = [Symbol(nameof(activation isa NamedTuple ? activation[k] : activation)) for k in neural]
    all(a -> haskey(ACT, a), acts) || throw(ArgumentError("unsupported: activation without a device implementation in $(acts)"))
    preds = [collect(Symbol, predictors[k]) for k in neural]
    all(!isempty, preds) || throw(ArgumentError("unsupported: a network without predictors"))
    flat = reduce(vcat, preds)
    # widths of the block-diagonal envelope; a shallower net rides identity blocks as wide as its last hidden layer
    tot = [sum(h[min(l, length(h))] for h in hl) for l in 1:maximum(length.(hl))]
    dims = [length(flat); tot; length(neural)]
    NN = [(dims[i + 1], dims[i]) for i in 1:(length(dims) - 1)]
    glob = collect(Symbol, global_param_names)
    fixed = [n for n in all_names if !(n in neural) && !(n in glob)]
    config = (; hidden_layers, activation = (length(unique(acts)) == 1 ? acts[1] : NamedTuple{Tuple(neural)}(Tuple(acts))),
        scale_nn_outputs, input_batchnorm, start_from_default, multi = (; predictors = preds, hidden = hl, activations = acts), kwargs...)
    return SingleNNHybridModel(NN, flat, collect(Symbol, forcing), collect(Symbol, targets), mechanistic_model, parameters,
        neural, glob, fixed, scale_nn_outputs, start_from_default, config)
end
const MultiNNHybridModel = SingleNNHybridModel      # one struct serves both; `haskey(m.config, :multi)` marks the multi-network form

function n_theta(m::SingleNNHybridModel)
    haskey(m.config, :multi) || return sum(o * i + o for (o, i) in m.NN) + length(m.global_param_names)
    n = 0
    for (p, h) in zip(m.config.multi.predictors, m.config.multi.hidden)     # theta holds the nets one after the other
        d = [length(p); h; 1]
        n += sum(d[i + 1] * d[i] + d[i + 1] for i in 1:(length(d) - 1))
    end
    return n + length(m.global_param_names)
end

pad(v, n, T) = ntuple(i -> i <= length(v) ? T(v[i]) : zero(T), n)

function descriptor(m::SingleNNHybridModel; device::Integer = 0)
    ms = MECH[m.mechanistic_model]
    pg = ms.program
    kind = Int32[]; index = Int32[]; def = Float32[]; lo = Float32[]; hi = Float32[]
    for p in ms.params
        if p in m.neural_param_names
            push!(kind, 0); push!(index, findfirst(==(p), m.neural_param_names) - 1)
        elseif p in m.global_param_names
            push!(kind, 1); push!(index, findfirst(==(p), m.global_param_names) - 1)
        else
            push!(kind, 2); push!(index, 0)
        end
        d, l, u = m.parameters[p]
        push!(def, d); push!(lo, l); push!(hi, u)
    end
    hidden = [o for (o, _) in m.NN[1:(end - 1)]]
    n_nets = Int32(0); net_p = Int32[]; net_h = zeros(Int32, 64); net_a = Int32[]; net_d = Int32[]
    act = m.config.activation isa Symbol ? ACT[m.config.activation] : 5          # 5 = EH_ACT_PER_NET
    if haskey(m.config, :multi)
        mu = m.config.multi
        length(mu.hidden) <= 8 || throw(ArgumentError("unsupported: more than 8 networks"))
        n_nets = Int32(length(mu.hidden)); net_p = Int32.(length.(mu.predictors)); net_a = Int32[ACT[a] for a in mu.activations]
        net_d = Int32.(length.(mu.hidden))
        for (k, h) in enumerate(mu.hidden), (l, w) in enumerate(h)
            net_h[(k - 1) * 8 + l] = w                                           # int32_t net_hidden[8][8], row-major
        end
    end
    return EhModelDesc(sizeof(EhModelDesc), device, length(m.predictors), length(hidden), pad(hidden, 8, Int32),
        act, m.scale_nn_outputs, m.config.input_batchnorm, ms.id, length(ms.params),
        pad(kind, 8, Int32), pad(index, 8, Int32), pad(def, 8, Float32), pad(lo, 8, Float32), pad(hi, 8, Float32),
        length(m.forcing), pad([findfirst(==(f), m.forcing) - 1 for f in ms.forcings], 4, Int32),
        length(m.targets), pad([findfirst(==(t), ms.outputs) - 1 for t in m.targets], 4, Int32),
        n_nets, pad(net_p, 8, Int32), pad(net_h, 64, Int32), pad(net_a, 8, Int32), pad(net_d, 8, Int32),    # MultiNN form (n_nets = 0: SingleNN)
        (pg === nothing ? (Int32(0), Int32(0), Int32(0), Int32(0), pad(Int32[], 3, Int32), pad(UInt32[], 64, UInt32), pad(Float32[], 16, Float32)) :
         (Int32(length(pg.code)), Int32(length(pg.consts)), Int32(length(ms.forcings)), Int32(length(pg.out)), pad(pg.out, 3, Int32),
          pad(pg.code, 64, UInt32), pad(pg.consts, 16, Float32)))...)
end

# ------------------------------------------------------------------------------------------------
# engine = one eh_handle
# ------------------------------------------------------------------------------------------------
mutable struct HybridEngine
    h::Ptr{Cvoid}
    model::SingleNNHybridModel
    n_theta::Int
end

function check(e::HybridEngine, st::Integer)
    st == 0 && return
    msg = unsafe_string(@ccall LIB[].eh_last_error(e.h::Ptr{Cvoid})::Cstring)
    st == -1 && throw(ArgumentError(msg))                   # EH_EINVAL      (reference: ArgumentError / AssertionError)
    st == -4 && throw(ArgumentError("unsupported: " * msg)) # EH_EUNSUPPORTED
    st == -3 && throw(OutOfMemoryError())
    error(msg)                                              # EH_EHIP / EH_ESTATE
end

function HybridEngine(m::SingleNNHybridModel; device::Integer = 0)
    d = Ref(descriptor(m; device))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    st = @ccall LIB[].eh_create(d::Ref{EhModelDesc}, h::Ref{Ptr{Cvoid}})::Int32
    st == 0 || error(unsafe_string(@ccall LIB[].eh_last_error(C_NULL::Ptr{Cvoid})::Cstring))
    e = HybridEngine(h[], m, n_theta(m))
    finalizer(x -> (@ccall LIB[].eh_destroy(x.h::Ptr{Cvoid})::Int32), e)
    return e
end

"X is (P × N) exactly as prepare_data returns it (src/data/prepare_data.jl:6); forcings / targets NamedTuples of Vector{Float32}."
function set_data!(e::HybridEngine, split::Integer, X::Matrix{Float32}, forcings::NamedTuple, targets::NamedTuple)
    f = [forcings[k] for k in e.model.forcing]; t = [targets[k] for k in e.model.targets]
    GC.@preserve X f t begin
        fp = [pointer(v) for v in f]; tp = [pointer(v) for v in t]
        check(e, @ccall LIB[].eh_set_data(e.h::Ptr{Cvoid}, split::Int32, size(X, 2)::Int64, X::Ptr{Float32},
            fp::Ptr{Ptr{Float32}}, tp::Ptr{Ptr{Float32}}, 0::Int32)::Int32)
    end
end
set_params!(e::HybridEngine, θ::Vector{Float32}) = check(e, @ccall LIB[].eh_set_params(e.h::Ptr{Cvoid}, θ::Ptr{Float32}, length(θ)::Int64)::Int32)
function get_params(e::HybridEngine)
    θ = Vector{Float32}(undef, e.n_theta)
    check(e, @ccall LIB[].eh_get_params(e.h::Ptr{Cvoid}, θ::Ptr{Float32}, length(θ)::Int64)::Int32)
    return θ
end
opt_init!(e::HybridEngine; rule = 0, eta = 0.01f0, beta = (0.9f0, 0.999f0), epsilon = 1.0f-8, lambda = 0.0f0) =
    check(e, @ccall LIB[].eh_opt_init(e.h::Ptr{Cvoid}, rule::Int32, eta::Float32, beta[1]::Float32, beta[2]::Float32, epsilon::Float32, lambda::Float32)::Int32)

"engine options: :max_blocks, :variant, :fast_paths, :row_split, :fused_update, :training_loss (0 mse, 1 rmse, 2 mae, 3 nseLoss, 4 pearsonLoss, 5 kgeLoss, 6 pbkgeLoss),
:specialize (1 = step kernels compiled at run time around this model's descriptor, about a second, ~20 % faster small-model steps),
:jit (recorded closures: 0 = interpret the program instead of compiling it)"
const LOSS_KINDS = (; mse = 0, rmse = 1, mae = 2, nseLoss = 3, pearsonLoss = 4, kgeLoss = 5, pbkgeLoss = 6)
"PerTarget((:mse, :mae)): one training loss per target, summed (src/losses/compute_loss.jl:128-145)"
struct PerTarget{T <: Tuple}
    losses::T
end
"training_loss = :mae (all targets alike) | PerTarget((...)) / a tuple or vector of symbols (one per target; mse / mae / nseLoss)"
function set_training_loss!(e::HybridEngine, spec)
    spec isa Symbol && return set_option!(e, :training_loss, LOSS_KINDS[spec])
    kinds = Int32[LOSS_KINDS[k] for k in (spec isa PerTarget ? spec.losses : spec)]
    check(e, @ccall LIB[].eh_set_target_losses(e.h::Ptr{Cvoid}, kinds::Ptr{Int32}, length(kinds)::Int32)::Int32)
end
set_option!(e::HybridEngine, name::Symbol, value::Integer) =
    check(e, @ccall LIB[].eh_set_option(e.h::Ptr{Cvoid}, String(name)::Cstring, value::Int64)::Int32)
"""
    set_agg!(e, agg; extra_terms = 0)

`agg::Function` of the training configuration (`src/config/TrainingConfig.jl:76-77`): the training loss is `agg(per-target losses)`
(`src/losses/compute_loss.jl:50-53`) and, with an extra loss, `agg([that, extra entries...])` (`:31-34`).  `sum` or `mean`
(`Statistics.mean`); `extra_terms` = the entries the extra loss returns (only `mean` needs it).
"""
function set_agg!(e::HybridEngine, agg; extra_terms::Integer = 0)
    name = Symbol(agg)
    name in (:sum, :mean) || throw(ArgumentError("agg = $agg: the device implements sum and mean"))
    set_option!(e, :extra_terms, extra_terms)
    set_option!(e, :agg, name === :mean ? 1 : 0)
end

"""
    set_weight_l2!(e, λ; normalize = false)

`extra_loss = (ŷ, ps) -> (; l2 = λ * weight_l2(ps; normalize))` (`src/utils/extract_weights.jl:69-91`, added through `agg = sum`,
`src/losses/compute_loss.jl:31-34`); `λ = 0` switches it off.
"""
set_weight_l2!(e::HybridEngine, λ::Real; normalize::Bool = false) =
    check(e, @ccall LIB[].eh_set_weight_l2(e.h::Ptr{Cvoid}, Float32(λ)::Float32, Int32(normalize)::Int32)::Int32)

"""
    set_weight_l2_coef!(e, coef)

Several `weight_l2` terms as one coefficient per flat-θ entry: extra loss `Σᵢ coef[i] θᵢ²` — e.g. the reference's own example
`(; l2_Rb = λ * weight_l2(ps.Rb; normalize = true),)` (`src/utils/extract_weights.jl:64`) is `λ / length(weights of net Rb)` at the
entries of that network's `weight` leaves and zero elsewhere (`l2_coefficients`).  `nothing` switches the extra loss off.
"""
function set_weight_l2_coef!(e::HybridEngine, coef::Union{Nothing, AbstractVector{<:Real}})
    coef === nothing && return check(e, @ccall LIB[].eh_set_weight_l2_coef(e.h::Ptr{Cvoid}, C_NULL::Ptr{Float32}, 0::Int64)::Int32)
    c = Vector{Float32}(coef)
    GC.@preserve c check(e, @ccall LIB[].eh_set_weight_l2_coef(e.h::Ptr{Cvoid}, c::Ptr{Float32}, length(c)::Int64)::Int32)
end

"(kernel pairs compiled at run time and in use, compiler / failure log) -- 0 with a log = the kernels built ahead of time run instead"
function jit_status(e::HybridEngine)
    n = Ref{Int32}(0); buf = Vector{UInt8}(undef, 8192)
    check(e, @ccall LIB[].eh_jit_status(e.h::Ptr{Cvoid}, n::Ptr{Int32}, buf::Ptr{UInt8}, length(buf)::Int64)::Int32)
    return Int(n[]), unsafe_string(pointer(buf))
end
synchronize(e::HybridEngine) = check(e, @ccall LIB[].eh_synchronize(e.h::Ptr{Cvoid})::Int32)

# data-parallel seam (one process per GPU, or one task per device): the gradient sums are all-reduced by the library itself
# (comm_init! / dp_allreduce! / dp_train_step! below: RCCL inside libeasyhybrid_hip.so) or by the caller's own collective on the device buffers
const EH_BUF_GRAD, EH_BUF_GACC, EH_BUF_BNSTAT = Int32(0), Int32(4), Int32(5)
function device_buffer(e::HybridEngine, which::Integer)
    p = Ref{Ptr{Cvoid}}(C_NULL); n = Ref{Int64}(0)
    check(e, @ccall LIB[].eh_device_buffer(e.h::Ptr{Cvoid}, which::Int32, p::Ref{Ptr{Cvoid}}, n::Ref{Int64})::Int32)
    return Ptr{Float32}(p[]), n[]
end
dp_grad!(e::HybridEngine, first::Integer, count::Integer) = check(e, @ccall LIB[].eh_dp_grad(e.h::Ptr{Cvoid}, first::Int64, count::Int64)::Int32)
"multi-target models under DP: this shard's per-target sums into EH_BUF_TCOUNT (buffer 6; all-reduce its 12 floats, then dp_grad!); dp_train_step! does it itself"
dp_counts!(e::HybridEngine, first::Integer, count::Integer) = check(e, @ccall LIB[].eh_dp_counts(e.h::Ptr{Cvoid}, first::Int64, count::Int64)::Int32)
"common shift of the shifted target sums: every rank passes the same vector (e.g. the global mean of each target) after set_data!"
set_target_shift!(e::HybridEngine, shift::Vector{Float32}; split = EH_SPLIT_TRAIN) =
    check(e, @ccall LIB[].eh_set_target_shift(e.h::Ptr{Cvoid}, split::Int32, shift::Ptr{Float32}, length(shift)::Int64)::Int32)
function dp_apply!(e::HybridEngine)
    loss = Ref{Float32}(NaN32)
    check(e, @ccall LIB[].eh_dp_apply(e.h::Ptr{Cvoid}, loss::Ref{Float32})::Int32)
    return loss[]
end
function dp_fused_step!(e::HybridEngine, first::Integer, count::Integer)
    k = Ref{Int32}(0)
    check(e, @ccall LIB[].eh_dp_fused_step(e.h::Ptr{Cvoid}, first::Int64, count::Int64, k::Ref{Int32})::Int32)
    return k[]          # which third of EH_BUF_GACC to all-reduce
end
# peer-to-peer exchange of the fused step (no collective per step): p2p_init! -> all-gather the handles (MPI) -> p2p_attach! -> p2p_selftest
function p2p_init!(e::HybridEngine, world::Integer, rank::Integer)
    hd = zeros(UInt8, 64)
    check(e, @ccall LIB[].eh_p2p_init(e.h::Ptr{Cvoid}, world::Int32, rank::Int32, hd::Ptr{UInt8}, 64::Int64)::Int32)
    return hd
end
p2p_attach!(e::HybridEngine, handles::Matrix{UInt8}) =            # 64 x world
    check(e, @ccall LIB[].eh_p2p_attach(e.h::Ptr{Cvoid}, handles::Ptr{UInt8}, 64::Int64)::Int32)
function p2p_selftest(e::HybridEngine; rounds::Integer = 8)
    ok = Ref{Int32}(0)
    check(e, @ccall LIB[].eh_p2p_selftest(e.h::Ptr{Cvoid}, rounds::Int32, ok::Ref{Int32})::Int32)
    return ok[] != 0
end
p2p_disable!(e::HybridEngine) = check(e, @ccall LIB[].eh_p2p_disable(e.h::Ptr{Cvoid})::Int32)

"the library's own RCCL communicator (include/easyhybrid_hip.h, eh_comm_*): data parallelism without NCCL.jl / MPI.jl"
function comm_unique_id()
    id = zeros(UInt8, 128)
    st = @ccall LIB[].eh_comm_unique_id(id::Ptr{UInt8}, 128::Int64)::Int32
    st == 0 || error("eh_comm_unique_id: status $st")
    return id
end
comm_init!(e::HybridEngine, id::Vector{UInt8}, world::Integer, rank::Integer) =
    check(e, @ccall LIB[].eh_comm_init(e.h::Ptr{Cvoid}, id::Ptr{UInt8}, length(id)::Int64, world::Int32, rank::Int32)::Int32)
comm_destroy!(e::HybridEngine) = check(e, @ccall LIB[].eh_comm_destroy(e.h::Ptr{Cvoid})::Int32)
"SUM all-reduce, in stream order, of EH_BUF_GRAD (0), a third of EH_BUF_GACC (4, `index`) or EH_BUF_BNSTAT (5)"
dp_allreduce!(e::HybridEngine, which::Integer, index::Integer = 0) = check(e, @ccall LIB[].eh_dp_allreduce(e.h::Ptr{Cvoid}, which::Int32, index::Int32)::Int32)
"one data-parallel step of this rank on its shard window (statistics + gradient exchange + update); every rank calls it"
dp_train_step!(e::HybridEngine, first::Integer, count::Integer) =
    check(e, @ccall LIB[].eh_dp_train_step(e.h::Ptr{Cvoid}, first::Int64, count::Int64, C_NULL::Ptr{Float32})::Int32)

"""
ONE Julia process driving several engines (one per device): `comm_init_local!(engines)` makes them a local group whose
all-reduces run inside the library without RCCL (events + one small kernel per member over peer-mapped memory; rank = position),
`dp_train_step_group!(engines, firsts, count)` is a whole data-parallel step of all of them from this one task.
"""
function comm_init_local!(es::Vector{HybridEngine})
    hs = Ptr{Cvoid}[e.h for e in es]
    GC.@preserve hs check(es[1], @ccall LIB[].eh_comm_init_local(hs::Ptr{Ptr{Cvoid}}, length(hs)::Int32)::Int32)
end
"""
The fused step kernels of the engines of ONE Julia process exchange their sums themselves (plain pointers to each other's
receive buffers, no IPC, no collective call per step).  The engines need `fused_update` on and a communicator for the
fall-back (`comm_init_local!`).  `p2p_init_local!` returns false when the start-up self-test failed (the engines then keep
all-reducing); `p2p_check_local!` drains all members and returns false after a missed exchange -- every member has then left the
exchange and taken member 1's parameters and optimiser state.
"""
function p2p_init_local!(es::Vector{HybridEngine}; rounds::Integer = 8)
    hs = Ptr{Cvoid}[e.h for e in es]; ok = Ref{Int32}(0)
    GC.@preserve hs check(es[1], @ccall LIB[].eh_p2p_init_local(hs::Ptr{Ptr{Cvoid}}, length(hs)::Int32, rounds::Int32, ok::Ref{Int32})::Int32)
    return ok[] != 0
end
function p2p_check_local!(es::Vector{HybridEngine})
    hs = Ptr{Cvoid}[e.h for e in es]; ok = Ref{Int32}(0)
    GC.@preserve hs check(es[1], @ccall LIB[].eh_p2p_check_local(hs::Ptr{Ptr{Cvoid}}, length(hs)::Int32, ok::Ref{Int32})::Int32)
    return ok[] != 0
end
comm_group_begin() = (st = @ccall LIB[].eh_comm_group_begin()::Int32; st == 0 || error("eh_comm_group_begin: status $st"); nothing)
comm_group_end() = (st = @ccall LIB[].eh_comm_group_end()::Int32; st == 0 || error("eh_comm_group_end: status $st"); nothing)
function dp_train_step_group!(es::Vector{HybridEngine}, firsts::Vector{<:Integer}, count::Integer)
    hs = Ptr{Cvoid}[e.h for e in es]; fs = Int64.(firsts)
    GC.@preserve hs fs check(es[1], @ccall LIB[].eh_dp_train_step_group(hs::Ptr{Ptr{Cvoid}}, length(hs)::Int32, fs::Ptr{Int64}, count::Int64, C_NULL::Ptr{Float32})::Int32)
end

"running (mean, var) of the input BatchNorm layer -- the `st.st_nn` part of the model state (`src/models/NNModels.jl:89-105`)"
function get_bn_state(e::HybridEngine, P::Integer)
    m = zeros(Float32, P); v = zeros(Float32, P)
    check(e, @ccall LIB[].eh_get_bn_state(e.h::Ptr{Cvoid}, m::Ptr{Float32}, v::Ptr{Float32}, P::Int64)::Int32)
    return m, v
end
set_bn_state!(e::HybridEngine, m::Vector{Float32}, v::Vector{Float32}) =
    check(e, @ccall LIB[].eh_set_bn_state(e.h::Ptr{Cvoid}, m::Ptr{Float32}, v::Ptr{Float32}, length(m)::Int64)::Int32)

"""
two-pass training losses (pearsonLoss / kgeLoss / pbkgeLoss; rmse on a multi-target model) under DP: the shard's moment sums of the
window into EH_BUF_MOMENT -- `stage = 0` about the common target shift, `stage = 1` (after the all-reduce) about the global mean of the
predictions; all-reduce after each, then `dp_grad!` (`eh_dp_train_step` / `dp_train_step_group!` do all of it inside the library)
"""
dp_moments!(e::HybridEngine, first::Integer, count::Integer, stage::Integer) =
    check(e, @ccall LIB[].eh_dp_moments(e.h::Ptr{Cvoid}, first::Int64, count::Int64, stage::Int32)::Int32)

"input BatchNorm under DP: shard sums into EH_BUF_BNSTAT (all-reduce it before dp_grad! / dp_fused_step!)"
set_bn_shift!(e::HybridEngine, c::Vector{Float32}) = check(e, @ccall LIB[].eh_set_bn_shift(e.h::Ptr{Cvoid}, c::Ptr{Float32}, length(c)::Int64)::Int32)
dp_bn_stats!(e::HybridEngine, first::Integer, count::Integer) = check(e, @ccall LIB[].eh_dp_bn_stats(e.h::Ptr{Cvoid}, first::Int64, count::Int64)::Int32)

"one Lux.Training.single_train_step! (src/training/epoch.jl:20-26) on train samples first+1 : first+count"
function train_step!(e::HybridEngine, first::Integer, count::Integer)
    loss = Ref{Float32}(NaN32)
    check(e, @ccall LIB[].eh_train_step(e.h::Ptr{Cvoid}, C_NULL::Ptr{Int32}, 0::Int32, first::Int64, count::Int64, loss::Ref{Float32})::Int32)
    return loss[]
end
"the same step on the minibatch an MLUtils.DataLoader drew (src/data/loaders.jl:1-12): `batch` = 1-based sample indices of the train split"
function train_step!(e::HybridEngine, batch::AbstractVector{<:Integer})
    idx = Int32.(batch .- 1)
    loss = Ref{Float32}(NaN32)
    GC.@preserve idx check(e, @ccall LIB[].eh_train_step(e.h::Ptr{Cvoid}, idx::Ptr{Int32}, 0::Int32, 0::Int64, length(idx)::Int64, loss::Ref{Float32})::Int32)
    return loss[]
end
"one run_epoch! (src/training/epoch.jl:13-33)"
function train_epoch!(e::HybridEngine, batchsize::Integer; seed::Integer = 0, shuffle::Bool = true)
    loss = Ref{Float32}(NaN32); n = Ref{Int64}(0)
    check(e, @ccall LIB[].eh_train_epoch(e.h::Ptr{Cvoid}, batchsize::Int64, seed::UInt64, shuffle::Int32, loss::Ref{Float32}, n::Ref{Int64})::Int32)
    return loss[], n[]
end
function loss_and_grad(e::HybridEngine, split::Integer, first::Integer, count::Integer)
    loss = Ref{Float32}(NaN32); nv = Ref{Int64}(0); g = Vector{Float32}(undef, e.n_theta)
    check(e, @ccall LIB[].eh_loss_and_grad(e.h::Ptr{Cvoid}, split::Int32, C_NULL::Ptr{Int32}, first::Int64, count::Int64,
        loss::Ref{Float32}, g::Ptr{Float32}, nv::Ref{Int64})::Int32)
    return loss[], g, nv[]
end
"evaluate_acc (src/training/train.jl:347-355): nested (mse = (reco = .., sum = ..), r2 = ...) like compute_loss.jl:55-66"
"""
    mech_loss_vjp!(e, o, forcings, targets, ∂o; n_valid = nothing) -> (loss, ∇globals, n_valid)

The mechanistic stage on its own (`eh_mech_loss_vjp`): `o` = outputs of a neural network evaluated elsewhere on the same GPU,
`B × K` (column k = neural parameter k), `forcings` / `targets` vectors of B — all DEVICE arrays (anything `pointer` returns a
device address for, e.g. `ROCArray{Float32}`); `∂o` receives d loss / d o.  Feed it to the network's pullback.
"""
function mech_loss_vjp!(e::HybridEngine, o, forcings, targets, ∂o; n_valid = nothing)
    B = size(o, 1)
    fp = Ptr{Float32}[reinterpret(Ptr{Float32}, pointer(f)) for f in forcings]; tp = Ptr{Float32}[reinterpret(Ptr{Float32}, pointer(t)) for t in targets]
    loss = Ref{Float32}(NaN32); nv = Ref{Int64}(0); g = zeros(Float32, max(1, length(e.model.global_param_names)))
    nvec = n_valid === nothing ? Int64[] : Int64.(collect(n_valid))
    nin = n_valid === nothing ? Ptr{Int64}(C_NULL) : pointer(nvec)
    GC.@preserve o ∂o forcings targets fp tp nvec check(e, @ccall LIB[].eh_mech_loss_vjp(e.h::Ptr{Cvoid}, B::Int64, B::Int64,
        reinterpret(Ptr{Float32}, pointer(o))::Ptr{Float32}, fp::Ptr{Ptr{Float32}}, tp::Ptr{Ptr{Float32}}, nin::Ptr{Int64},
        reinterpret(Ptr{Float32}, pointer(∂o))::Ptr{Float32}, C_NULL::Ptr{Float32}, loss::Ref{Float32}, g::Ptr{Float32}, nv::Ref{Int64})::Int32)
    return loss[], g[1:length(e.model.global_param_names)], nv[]
end

function evaluate(e::HybridEngine, split::Integer, n::Integer; loss_types = [:mse, :r2], agg = sum)
    T = length(e.model.targets)
    m = Vector{EhTargetMetrics}(undef, T)
    check(e, @ccall LIB[].eh_eval(e.h::Ptr{Cvoid}, split::Int32, 0::Int64, n::Int64, m::Ptr{EhTargetMetrics}, C_NULL::Ptr{Ptr{Float32}}, C_NULL::Ptr{Ptr{Float32}})::Int32)
    return NamedTuple{Tuple(loss_types)}(map(loss_types) do lt
        per = [getfield(m[t], lt) for t in 1:T]
        NamedTuple{(e.model.targets..., Symbol(agg))}((per..., Symbol(agg) === :mean ? sum(per) / length(per) : sum(per)))      # compute_loss.jl:55-66
    end)
end
function forward(e::HybridEngine, split::Integer, n::Integer)
    ŷ = [Vector{Float32}(undef, n) for _ in e.model.targets]
    GC.@preserve ŷ begin
        p = [pointer(v) for v in ŷ]
        check(e, @ccall LIB[].eh_forward(e.h::Ptr{Cvoid}, split::Int32, 0::Int64, n::Int64, p::Ptr{Ptr{Float32}}, C_NULL::Ptr{Ptr{Float32}})::Int32)
    end
    return NamedTuple{Tuple(e.model.targets)}(Tuple(ŷ))
end

# ------------------------------------------------------------------------------------------------
# train: the epoch loop of _train (src/training/train.jl:95-136) with the hot path on the device
# ------------------------------------------------------------------------------------------------
"""
    train!(engine, ((x_train, forcings_train), y_train), ((x_val, forcings_val), y_val); nepochs, batchsize, ...)

In-place variant: updates the parameters held by `engine`.  `train(model, data...; kwargs...)`
builds an engine, calls this and returns `(; ps, train_history, val_history, best_epoch, best_loss)`.
"""
function train!(e::HybridEngine, train_data, val_data; nepochs = 200, batchsize = 64, eta = 0.01f0, patience = typemax(Int),
        loss_types = [:mse, :r2], random_seed = 161803, return_model = :best)
    ((xt, ft), yt), ((xv, fv), yv) = train_data, val_data
    set_data!(e, EH_SPLIT_TRAIN, xt, ft, yt); set_data!(e, EH_SPLIT_VAL, xv, fv, yv)
    opt_init!(e; eta)
    nt, nv = size(xt, 2), size(xv, 2)
    hist_t = Any[evaluate(e, EH_SPLIT_TRAIN, nt; loss_types)]; hist_v = Any[evaluate(e, EH_SPLIT_VAL, nv; loss_types)]
    best_loss = hist_v[1][1].sum; best_ps = get_params(e); best_epoch = 0; counter = 0
    has_bn = get(e.model.config, :input_batchnorm, false) === true
    best_bn = has_bn ? get_bn_state(e, length(e.model.predictors)) : nothing                         # the running statistics belong to the epoch's model state
    better = first(loss_types) in (:pearson, :r2, :nse, :kge) ? (>) : (<)        # loss_fn.jl:181-194
    for epoch in 1:nepochs
        train_epoch!(e, batchsize; seed = random_seed + epoch, shuffle = true)   # run_epoch!
        push!(hist_t, evaluate(e, EH_SPLIT_TRAIN, nt; loss_types)); push!(hist_v, evaluate(e, EH_SPLIT_VAL, nv; loss_types))   # evaluate_epoch
        cur = hist_v[end][1].sum
        if better(cur, best_loss)
            best_loss, best_ps, best_epoch, counter = cur, get_params(e), epoch, 0
            has_bn && (best_bn = get_bn_state(e, length(e.model.predictors)))
        else
            counter += 1
        end
        counter >= patience && break
    end
    if return_model == :best
        set_params!(e, best_ps)
        has_bn && set_bn_state!(e, best_bn...)
    end
    return (; ps = get_params(e), train_history = hist_t, val_history = hist_v, best_epoch, best_loss)
end

function train(m::SingleNNHybridModel, train_data, val_data; θ0::Vector{Float32}, device = 0, kwargs...)
    e = HybridEngine(m; device)
    set_params!(e, θ0)
    return train!(e, train_data, val_data; kwargs...)
end

# ------------------------------------------------------------------------------------------------
# the reference's front door: train(model, data; kwargs...)  (src/training/train.jl:211-219)
# ------------------------------------------------------------------------------------------------
"Optimisers.jl-style rules the device implements (EasyHybrid.jl:59 re-exports the originals; pass those or these)"
struct Adam; eta::Float32; beta::Tuple{Float32, Float32}; epsilon::Float32; end
Adam(eta = 0.001f0, beta = (0.9f0, 0.999f0)) = Adam(eta, beta, 1.0f-8)
struct AdamW; eta::Float32; beta::Tuple{Float32, Float32}; lambda::Float32; epsilon::Float32; end
AdamW(eta = 0.001f0, beta = (0.9f0, 0.999f0), lambda = 0.0f0) = AdamW(eta, beta, lambda, 1.0f-8)
struct RMSProp; eta::Float32; rho::Float32; epsilon::Float32; end
RMSProp(eta = 0.001f0, rho = 0.9f0) = RMSProp(eta, rho, 1.0f-8)
struct Descent; eta::Float32; end
# (duck-typed on the field names, so Optimisers.Adam(0.01) etc. work as well)
function _opt_args(o)
    n = nameof(typeof(o))
    n == :Adam && return (; rule = 0, eta = Float32(o.eta), beta = Float32.(o.beta), epsilon = Float32(o.epsilon), lambda = 0.0f0)
    n == :AdamW && return (; rule = 1, eta = Float32(o.eta), beta = Float32.(o.beta), epsilon = Float32(o.epsilon), lambda = Float32(o.lambda))
    n == :RMSProp && return (; rule = 2, eta = Float32(o.eta), beta = (Float32(o.rho), 0.0f0), epsilon = Float32(o.epsilon), lambda = 0.0f0)
    n == :Descent && return (; rule = 3, eta = Float32(o.eta), beta = (0.0f0, 0.0f0), epsilon = 0.0f0, lambda = 0.0f0)
    throw(ArgumentError("optimiser $(typeof(o)): the device runs Adam / AdamW / RMSProp / Descent"))
end

_col(data, n::Symbol) = Float32.(collect(data isa AbstractDict ? data[n] : getproperty(data, n)))   # NamedTuple / Dict of columns, DataFrame, ...

"""
    prepare_data(model, data) -> ((X, forcings), targets)

`src/data/prepare_data.jl:6-60`: the columns the model names, as Float32 -- predictors as a (P x N) matrix, forcings and targets
as NamedTuples of vectors; a row is kept only if its predictors and forcings are complete AND at least one target is present
(`src/data/prepare_data.jl:44-52`); the remaining missing targets stay NaN (they become the mask, `src/training/train.jl:221-232`).
"""
function prepare_data(m::SingleNNHybridModel, data)
    X = permutedims(reduce(hcat, [_col(data, p) for p in m.predictors]))
    F = [_col(data, f) for f in m.forcing]; Y = [_col(data, t) for t in m.targets]
    keep = vec(.!any(isnan, X; dims = 1))
    for f in F; keep .&= .!isnan.(f); end
    isempty(Y) || (keep .&= reduce((a, b) -> a .| b, [.!isnan.(y) for y in Y]))                  # prepare_data.jl:44-52: at least one target present
    return (Matrix{Float32}(X[:, keep]), NamedTuple{Tuple(m.forcing)}(Tuple(f[keep] for f in F))), NamedTuple{Tuple(m.targets)}(Tuple(y[keep] for y in Y))
end

"`src/data/split_data.jl:74-78` (the default branch): `splitobs(1:n; at = split_data_at, shuffle = shuffleobs)`"
function split_data(m::SingleNNHybridModel, data; split_data_at::Real = 0.8, shuffleobs::Bool = false, rng = Random.default_rng())
    (X, F), Y = prepare_data(m, data)
    n = size(X, 2)
    idx = shuffleobs ? Random.randperm(rng, n) : collect(1:n)
    ntr = round(Int, split_data_at * n)
    pick(ix) = ((X[:, ix], map(v -> v[ix], F)), map(v -> v[ix], Y))
    return pick(idx[1:ntr]), pick(idx[(ntr + 1):end])
end

"""
    initialparameters(rng, model) -> Vector{Float32}

Flat θ in the reference's ComponentArray order (`GenericHybridModel.jl:236-256`): per Dense layer the weight (column-major
`(out, in)`: `kaiming_uniform` with the activation's gain, Lux ≥ 1.0) and bias (`U(±1/√fan_in)`), then every global
parameter's raw value from its default (`start_from_default`, `:244-249`).  Julia's own RNG stream: not the reference's values.
"""
function initialparameters(rng, m::SingleNNHybridModel)
    act = get(m.config, :activation, :tanh)                  # (the constructor keeps the activation's name)
    gain = act === :tanh ? 5.0f0 / 3 : (act === :relu ? sqrt(2.0f0) : 1.0f0)
    θ = Float32[]
    for (li, (o, i)) in enumerate(m.NN)
        bw = (li < length(m.NN) ? gain : 1.0f0) * sqrt(3.0f0 / i)
        append!(θ, (2 .* rand(rng, Float32, o * i) .- 1) .* bw)
        append!(θ, (2 .* rand(rng, Float32, o) .- 1) ./ sqrt(Float32(i)))
    end
    for g in m.global_param_names
        d, lo, hi = m.parameters[g]
        push!(θ, m.start_from_default ? log((d - lo) / (hi - lo) / (1 - (d - lo) / (hi - lo))) : rand(rng, Float32))   # scale_single_param_minmax, :361-365
    end
    return θ
end

"""
    train(model, data; nepochs = 200, batchsize = 64, opt = Adam(0.01), patience, loss_types = [:mse, :r2],
          training_loss = :mse, random_seed = 161803, return_model = :best, split_data_at = 0.8, shuffleobs = false,
          train_from = nothing, device = 0)

The reference's `train(model, data; kwargs...)` (`src/training/train.jl:211-219` → `_train`, `:95-136`): prepare and split the
table, initial parameters, then per epoch one `run_epoch!` (`eh_train_epoch`: every minibatch of the shuffled train split on the
device) and one `evaluate_epoch` (`eh_eval` on both splits), early stopping on the first `loss_types` entry.  Returns the fields
of `TrainResults` the device path produces: `(; train_history, val_history, train_obs_pred, val_obs_pred, ps, st, best_epoch,
best_loss)`.  `data`: a DataFrame or a NamedTuple / Dict of equally long columns.
"""
function train(m::SingleNNHybridModel, data; nepochs = 200, batchsize = 64, opt = Adam(0.01f0), patience = typemax(Int),
        loss_types = [:mse, :r2], training_loss = :mse, random_seed = 161803, return_model = :best, split_data_at = 0.8,
        shuffleobs = false, train_from = nothing, device = 0, agg = sum)
    nepochs >= 0 || throw(ArgumentError("nepochs must be >= 0"))                               # validate_config, TrainingConfig.jl:162-180
    batchsize >= 1 || throw(ArgumentError("batchsize must be >= 1"))
    training_loss isa Function && throw(ArgumentError("this shim has no tracer for a custom loss function (the Python front door records one: program.trace_loss)"))
    rng = random_seed === nothing ? Random.default_rng() : Random.Xoshiro(random_seed)
    tr, va = split_data(m, data; split_data_at, shuffleobs, rng)
    size(tr[1][1], 2) == 0 && return nothing                                                    # train.jl:186
    e = HybridEngine(m; device)
    set_params!(e, train_from === nothing ? initialparameters(rng, m) : Float32.(train_from))
    o = _opt_args(opt)
    ((xt, ft), yt), ((xv, fv), yv) = tr, va
    set_data!(e, EH_SPLIT_TRAIN, xt, ft, yt); set_data!(e, EH_SPLIT_VAL, xv, fv, yv)
    opt_init!(e; rule = o.rule, eta = o.eta, beta = o.beta, epsilon = o.epsilon, lambda = o.lambda)
    set_training_loss!(e, training_loss)
    set_agg!(e, agg)
    aggn = Symbol(agg)
    nt, nv = size(xt, 2), size(xv, 2)
    hist_t = Any[evaluate(e, EH_SPLIT_TRAIN, nt; loss_types, agg)]; hist_v = Any[evaluate(e, EH_SPLIT_VAL, nv; loss_types, agg)]
    best_loss = getfield(hist_v[1][1], aggn); best_ps = get_params(e); best_epoch = 0; counter = 0
    has_bn = get(m.config, :input_batchnorm, false) === true
    best_bn = has_bn ? get_bn_state(e, length(m.predictors)) : nothing                         # the running statistics belong to the epoch's model state
    better = first(loss_types) in (:pearson, :r2, :nse, :kge) ? (>) : (<)                       # loss_fn.jl:181-194
    seed0 = random_seed === nothing ? rand(rng, UInt32) : random_seed
    for epoch in 1:nepochs
        train_epoch!(e, batchsize; seed = seed0 + epoch, shuffle = true)
        push!(hist_t, evaluate(e, EH_SPLIT_TRAIN, nt; loss_types, agg)); push!(hist_v, evaluate(e, EH_SPLIT_VAL, nv; loss_types, agg))
        cur = getfield(hist_v[end][1], aggn)
        if better(cur, best_loss)
            best_loss, best_ps, best_epoch, counter = cur, get_params(e), epoch, 0              # early_stopping.jl:16-42
            has_bn && (best_bn = get_bn_state(e, length(m.predictors)))
        else
            counter += 1
        end
        counter >= patience && break
    end
    if return_model == :best                                                                    # best_or_final: parameters AND state of that epoch
        set_params!(e, best_ps)
        has_bn && set_bn_state!(e, best_bn...)
    end
    obs_pred(split, y, n) = n == 0 ? (;) : merge(y, NamedTuple{Tuple(Symbol(t, :_pred) for t in m.targets)}(Tuple(values(forward(e, split, n)))))
    st = (; fixed = NamedTuple{Tuple(m.fixed_param_names)}(Tuple(Float32(m.parameters[f][1]) for f in m.fixed_param_names)))
    return (; train_history = hist_t, val_history = hist_v, train_obs_pred = obs_pred(EH_SPLIT_TRAIN, yt, nt),
            val_obs_pred = obs_pred(EH_SPLIT_VAL, yv, nv), ps = get_params(e), st, best_epoch, best_loss)
end

end # module
