"""BASELINE config 5: fp64 vs fp32-accumulate J'WJ on the step, the adjusted coordinates, diag Qxx and ||Qxx||_F.

Three assemblies of the same scene: 0 = structure-aware fp64 (the product path), 1 = densified A'(PA) on the fp64 matrix
cores, 2 = the same contraction with operands rounded to fp32 and fp32 MFMA accumulation.  Everything after the assembly
(factorisation, inverse) is fp64 in all three.  usage: precision_sweep.py [config] > profiles/...json"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bundle_adjustment_amd import engine, scene

name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
if name.startswith("mid"):        # 12 images x 150 points, dense per-image dispersions (U = 530)
    fp = scene.make_scene(12, 150, 90, dist=scene.DIST_FULL, weights="block", n_control=5, control_dense=True)
elif name == "cfg3_block":        # config 3's size (100 images x 1 000 points) with config 4's dense per-image dispersions (U = 3 614)
    fp = scene.make_scene(100, 1000, 400, dist=scene.DIST_FULL, weights="block", n_control=15, control_dense=True)
else:
    fp = scene.config(name)
s2, U = fp.sigma2apriori, fp.n_unknowns
P3 = 3 * fp.n_points


def run(mode):
    out = {}
    eng = engine.Engine(fp, assembly_mode=mode)
    try:
        eng.set_parameters(fp.values)
        eng.prepare_inverse(engine.INVERT_FULL)
        eng.build(s2, 0.0)
        out["first_step"] = eng.solve(False)
        steps = []
        for _ in range(6):
            eng.prepare_inverse(engine.INVERT_FULL)
            eng.build(s2, 0.0)
            dx = eng.solve(False)
            steps.append(eng.update(dx))
        out["steps"] = steps
        out["values"] = eng.get_parameters()
        eng.prepare_inverse(engine.INVERT_FULL)
        eng.build(s2, 0.0)
        dx = eng.solve(engine.INVERT_FULL)
        out["omega"] = eng.omega(s2, dx)
        Q = eng.get_cofactor()
        idx = np.arange(U)
        out["diagQ"] = Q[idx * (idx + 3) // 2].copy()
        off = 2.0 * np.dot(Q, Q) - np.dot(out["diagQ"], out["diagQ"])     # packed triangle -> full Frobenius norm
        out["normQ"] = float(np.sqrt(off))
    except engine.EngineError as ex:
        out["error"] = str(ex)
    finally:
        eng.close()
    return out


res = {m: run(m) for m in (0, 1, 2)}
ref = res[1]
report = {"config": name, "U": U, "reference": "assembly_mode 1 (densified contraction, fp64 operands and accumulation)", "modes": {}}
for m, label in ((0, "structure-aware fp64 (product path)"), (2, "densified, fp32 operands + fp32 MFMA accumulation")):
    r = res[m]
    if "error" in r or "error" in ref:
        report["modes"][label] = {"error": r.get("error") or ref.get("error")}
        continue
    d = {"first_step_max_abs_diff_over_max_step": float(np.abs(r["first_step"] - ref["first_step"]).max() / np.abs(ref["first_step"]).max()),
         "max_abs_dx_per_pass": [float(x) for x in r["steps"]],
         "adjusted_point_coordinates_max_abs_diff_mm": float(np.abs(r["values"][:P3] - ref["values"][:P3]).max()),
         "adjusted_parameters_max_rel_diff": float((np.abs(r["values"] - ref["values"]) / np.maximum(np.abs(ref["values"]), 1.0)).max()),
         "sigma0_ratio": float(r["omega"] / fp.degree_of_freedom / s2),
         "diag_Qxx_max_rel_diff": float((np.abs(r["diagQ"] - ref["diagQ"]) / np.abs(ref["diagQ"])).max()),
         "Qxx_frobenius_rel_diff": float(abs(r["normQ"] - ref["normQ"]) / ref["normQ"])}
    report["modes"][label] = d
report["reference_sigma0_ratio"] = float(ref["omega"] / fp.degree_of_freedom / s2) if "omega" in ref else None
report["reference_max_abs_dx_per_pass"] = [float(x) for x in ref.get("steps", [])]
print(json.dumps(report, indent=1))
