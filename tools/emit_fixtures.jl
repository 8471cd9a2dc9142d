# OPTIONAL, never a dependency, never shipped to the GPU box: for anyone who has Julia >= 1.10 with EasyHybrid.jl's packages
# installed (the build container has neither), this runs the small golden cases of tests/golden/csv_for_emit_fixtures/ (written by
# tools/export_case_csv.py from the committed .npz fixtures) through the REAL reference path -- constructHybridModel +
# Lux + Zygote + Optimisers.Adam -- and prints how far the committed oracle values are from what the reference computes.
# The oracle's gradients / Adam steps are otherwise "parity unpinned" by the reference's own tests (SURVEY.md section 8c).
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

root = joinpath(@__DIR__, "..", "tests", "golden", "csv_for_emit_fixtures")
println("relative deviation of the committed ORACLE values from the reference path (expected: ~1e-6, fp32 rounding):")
foreach(run_case, filter(isdir, readdir(root; join = true)))
