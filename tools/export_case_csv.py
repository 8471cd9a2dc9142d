"""Writes the small golden cases (tests/golden/rbq10_*_B12.npz) as plain CSV under tests/golden/csv_for_emit_fixtures/<case>/ so that
tools/emit_fixtures.jl can read them with Julia's standard library alone:  python tools/export_case_csv.py"""
import glob, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "rbq10_*_B12.npz"))):
    z = np.load(path)
    spec = json.loads(str(z["spec"]))
    name = os.path.basename(path)[:-4]
    out = os.path.join(ROOT, "tests", "golden", "csv_for_emit_fixtures", name)
    os.makedirs(out, exist_ok=True)
    np.savetxt(os.path.join(out, "theta.csv"), z["theta"][None].astype(np.float64), delimiter=",", fmt="%.9g")
    np.savetxt(os.path.join(out, "X.csv"), z["X"].astype(np.float64), delimiter=",", fmt="%.9g")                   # (P x B), as the reference holds it
    np.savetxt(os.path.join(out, "ta.csv"), z["forcing_ta"][None].astype(np.float64), delimiter=",", fmt="%.9g")
    np.savetxt(os.path.join(out, "reco.csv"), z["target_reco"][None].astype(np.float64), delimiter=",", fmt="%.9g")   # nan = missing
    np.savetxt(os.path.join(out, "expect_loss.csv"), np.array([[float(z["loss"])]]), delimiter=",", fmt="%.17g")
    np.savetxt(os.path.join(out, "expect_grad.csv"), z["grad"][None].astype(np.float64), delimiter=",", fmt="%.17g")
    np.savetxt(os.path.join(out, "expect_yhat.csv"), z["yhat_reco"][None].astype(np.float64), delimiter=",", fmt="%.17g")
    np.savetxt(os.path.join(out, "expect_theta_after_1.csv"), z["theta_after_1"][None].astype(np.float64), delimiter=",", fmt="%.9g")
    open(os.path.join(out, "spec.txt"), "w").write(f"activation={spec['activation']}\nscale_nn_outputs={str(spec['scale_nn_outputs']).lower()}\nhidden={','.join(map(str, spec['hidden']))}\n")
    print(name, "->", out)
