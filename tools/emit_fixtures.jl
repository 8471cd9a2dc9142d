# OPTIONAL, never a dependency, never shipped to the GPU box: for anyone who has Julia >= 1.10 with EasyHybrid.jl's packages
# installed (the build container has neither), this runs the small golden cases of tests/golden/csv_for_emit_fixtures/ (written by
# tools/export_case_csv.py from the committed .npz fixtures) through the REAL reference path -- constructHybridModel +
# Lux + Zygote + Optimisers.Adam -- and prints how far the committed oracle values are from what the reference computes.
# The oracle's gradients / Adam steps are otherwise "parity unpinned" by the reference's own tests (SURVEY.md section 8c).
# Round 6: the extended cases of tools/export_extended_cases.py (ext_*) -- input BatchNorm (train / test mode, running statistics), RMSProp /
# AdamW / Descent, nseLoss / pearsonLoss / kgeLoss, two targets under agg = mean, extra_loss = lambda * weight_l2, MultiNNHybridModel, a Chain
# with per-layer activations.
#
#   julia --project=/path/to/EasyHybrid.jl tools/emit_fixtures.jl
#
# Only Julia's standard library reads the files (DelimitedFiles); nothing here is used by the build or the tests.
using EasyHybrid, Lux, LuxCore, Zygote, Optimisers, ComponentArrays, Random, DelimitedFiles, Statistics

RbQ10(; ta, Q10, rb, tref = 15.0f0) = (; reco = rb .* Q10 .^ (0.1f0 .* (ta .- tref)), Q10, rb)      # test/test_split_data_train.jl:36-39
const PARAMS = (rb = (3.0f0, 0.0f0, 13.0f0), Q10 = (2.0f0, 1.0f0, 4.0f0))                            # :42-45

readrow(p) = Float32.(vec(readdlm(p, ',', Float64)))
relerr(a, b) = maximum(abs.(Float64.(a) .- Float64.(b))) / max(maximum(abs.(Float64.(b))), 1e-30)

function run_case(dir)
    spec = Dict(split(l, '=')[1] => split(l, '=')[2] for l in readlines(joinpath(dir, "spec.txt")))
    act = Dict("tanh" => tanh, "sigmoid" => Lux.sigmoid, "relu" => Lux.relu, "swish" => Lux.swish)[spec["activation"]]
    hidden = parse.(Int, split(spec["hidden"], ','))
    HM = constructHybridModel([:sw_pot, :dsw_pot], [:ta], [:reco], RbQ10, PARAMS, [:rb], [:Q10];
                              hidden_layers = hidden, activation = act, scale_nn_outputs = spec["scale_nn_outputs"] == "true")
    ps, st = LuxCore.setup(Random.default_rng(), HM)
    ps = ComponentArray(ps)
    theta = readrow(joinpath(dir, "theta.csv"))
    @assert length(theta) == length(ps) "flat parameter count differs: $(length(theta)) vs $(length(ps))"
    ps .= theta                                            # flat order: layer weights (column-major), biases, then the global raws (SURVEY.md a11)
    X = Float32.(readdlm(joinpath(dir, "X.csv"), ',', Float64))          # (P x B)
    ta, y = readrow(joinpath(dir, "ta.csv")), readrow(joinpath(dir, "reco.csv"))
    mask = .!isnan.(y)
    x = (X, (; ta))                                        # what prepare_data hands the model: (predictors, forcings)
    objective(p) = begin
        yhat, _ = HM(x, p, st)
        mean(abs2, yhat.reco[mask] .- y[mask])             # loss_fn(..., Val(:mse)), src/losses/loss_fn.jl:61-63, one target
    end
    loss, back = Zygote.pullback(objective, ps)
    grad = back(1.0f0)[1]
    yhat, _ = HM(x, ps, st)
    opt = Optimisers.setup(Optimisers.Adam(0.01f0), ps)
    _, ps1 = Optimisers.update(opt, ps, grad)
    println(rpad(basename(dir), 32),
            " loss ", relerr([loss], readrow(joinpath(dir, "expect_loss.csv"))),
            " yhat ", relerr(yhat.reco, readrow(joinpath(dir, "expect_yhat.csv"))),
            " grad ", relerr(collect(grad), readrow(joinpath(dir, "expect_grad.csv"))),
            " theta after one Adam step (abs) ", maximum(abs.(collect(ps1) .- readrow(joinpath(dir, "expect_theta_after_1.csv")))))
end

# ---- extended cases (tools/export_extended_cases.py): what rounds 2-5 added to the path -- input BatchNorm, the other optimiser rules, the
# moment losses, two targets under agg = mean, the weight_l2 extra loss, MultiNNHybridModel, a Chain with per-layer activations.
# spec.txt names what a case exercises; every expect_*.csv is an ORACLE value, printed here as its distance from the reference path.
reco2(; ta, Q10, rb, tref = 15.0f0) = (; reco = rb .* Q10 .^ (0.1f0 .* (ta .- tref)), half = 0.5f0 .* rb .+ 0.015625f0 .* ta, Q10, rb)
has(dir, f) = isfile(joinpath(dir, f))

function flat!(ps, theta)                 # the flat order of SURVEY.md a11: ComponentArray(ps) of the model, filled in order
    pc = ComponentArray(ps)
    @assert length(theta) == length(pc) "flat parameter count differs: $(length(theta)) vs $(length(pc))"
    pc .= theta
    return pc
end

function run_ext_case(dir)
    spec = Dict(split(l, '=')[1] => split(l, '=')[2] for l in readlines(joinpath(dir, "spec.txt")))
    acts = Dict("tanh" => tanh, "sigmoid" => Lux.sigmoid, "relu" => Lux.relu, "swish" => Lux.swish)
    act = acts[spec["activation"]]
    scale = get(spec, "scale_nn_outputs", "false") == "true"
    bnorm = get(spec, "input_batchnorm", "false") == "true"
    X = Float32.(readdlm(joinpath(dir, "X.csv"), ',', Float64))
    ta = readrow(joinpath(dir, "ta.csv"))
    targets = Symbol.(split(get(spec, "targets", "reco"), ','))
    ys = NamedTuple{Tuple(targets)}(Tuple(readrow(joinpath(dir, string(t) * ".csv")) for t in targets))
    masks = map(y -> .!isnan.(y), ys)
    model = spec["model"]
    if model == "rbq10_multinn"
        HM = constructHybridModel((rb = [:sw_pot, :dsw_pot], Q10 = [:p3]), [:ta], [:reco], RbQ10, PARAMS, Symbol[];
                                  hidden_layers = (rb = parse.(Int, split(spec["hidden_rb"], ',')), Q10 = parse.(Int, split(spec["hidden_Q10"], ','))),
                                  activation = act, scale_nn_outputs = scale)
        x = ((rb = X[1:2, :], Q10 = X[3:3, :]), (; ta))
    else
        hidden = parse.(Int, split(spec["hidden"], ','))
        hl = haskey(spec, "chain") ? Chain(Dense(hidden[1], hidden[2], Lux.relu)) : hidden          # chain=Dense(8,6,relu): the second hidden layer with an activation of its own
        mech = model == "reco2" ? reco2 : RbQ10
        HM = constructHybridModel([:sw_pot, :dsw_pot], [:ta], collect(targets), mech, PARAMS, [:rb], [:Q10];
                                  hidden_layers = hl, activation = act, scale_nn_outputs = scale, input_batchnorm = bnorm)
        x = (X, (; ta))
    end
    ps, st = LuxCore.setup(Random.default_rng(), HM)
    ps = flat!(ps, readrow(joinpath(dir, "theta.csv")))
    kind = Symbol(get(spec, "training_loss", "mse"))
    agg = get(spec, "agg", "sum") == "mean" ? mean : sum
    lam = parse(Float32, get(spec, "l2_lambda", "0"))
    objective(p) = begin
        yhat, st2 = HM(x, p, st)                            # train mode: with input BatchNorm the batch statistics normalise, st2 carries the running ones
        per = [EasyHybrid.loss_fn(getproperty(yhat, t), ys[t], masks[t], Val(kind)) for t in targets]
        l = agg(per)
        if get(spec, "extra", "") == "two_outputs"          # an entry over two outputs of the model and the RAW global parameter (compute_loss.jl:31-34)
            c = 0.05f0 * mean(yhat.reco .* yhat.half) * p.Q10[1] + 0.5f0 * mean((yhat.reco .- yhat.half) .^ 2)
            return agg([l, c])
        end
        lam > 0 ? agg([l, lam * EasyHybrid.weight_l2(p)]) : l      # compute_loss.jl:31-34: agg([loss, extra...])
    end
    loss, back = Zygote.pullback(objective, ps)
    grad = back(1.0f0)[1]
    out = rpad(basename(dir), 28) * " loss " * string(relerr([loss], readrow(joinpath(dir, "expect_loss.csv")))) *
          " grad " * string(relerr(collect(grad), readrow(joinpath(dir, "expect_grad.csv"))))
    if has(dir, "expect_yhat.csv")
        yhat, _ = HM(x, ps, st)
        out *= " yhat " * string(relerr(yhat.reco, readrow(joinpath(dir, "expect_yhat.csv"))))
    end
    if haskey(spec, "optimiser")
        lr = parse(Float32, spec["lr"])
        rule = spec["optimiser"] == "RMSProp" ? Optimisers.RMSProp(lr) : spec["optimiser"] == "AdamW" ? Optimisers.AdamW(lr, (0.9f0, 0.999f0), parse(Float32, spec["lambda"])) :
               spec["optimiser"] == "Descent" ? Optimisers.Descent(lr) : Optimisers.Adam(lr)
        _, ps1 = Optimisers.update(Optimisers.setup(rule, ps), ps, grad)
        out *= " theta after one " * spec["optimiser"] * " step (abs) " * string(maximum(abs.(collect(ps1) .- readrow(joinpath(dir, "expect_theta_after_1.csv")))))
        if bnorm
            _, st1 = HM(x, ps, st)                           # the state after the step's forward: running statistics advanced once
            bnst = st1.st_nn.layer_1                         # (InputBatchNorm is the chain's first layer; a wrapper layer: its state is the BatchNorm's)
            out *= " running mean " * string(relerr(vec(bnst.running_mean), readrow(joinpath(dir, "expect_running_mean_after_1.csv")))) *
                   " var " * string(relerr(vec(bnst.running_var), readrow(joinpath(dir, "expect_running_var_after_1.csv"))))
            yt, _ = HM(x, ps1, LuxCore.testmode(st1))
            out *= " test-mode yhat " * string(relerr(yt.reco, readrow(joinpath(dir, "expect_yhat_testmode_after_1.csv"))))
        end
    end
    println(out)
end

root = joinpath(@__DIR__, "..", "tests", "golden", "csv_for_emit_fixtures")
println("relative deviation of the committed ORACLE values from the reference path (expected: ~1e-6, fp32 rounding):")
dirs = filter(isdir, readdir(root; join = true))
foreach(run_case, filter(d -> !startswith(basename(d), "ext_"), dirs))
println("extended cases (BatchNorm, optimiser rules, moment losses, agg = mean, weight_l2, MultiNN, Chain with per-layer activations):")
for d in filter(d -> startswith(basename(d), "ext_"), dirs)
    try
        run_ext_case(d)
    catch e                                # (a case the installed EasyHybrid / Lux versions spell differently must not hide the others)
        println(rpad(basename(d), 28), " FAILED TO RUN: ", sprint(showerror, e))
    end
end
