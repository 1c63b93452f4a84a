"""Hunt for a flaky zero EO step (tests/test_gpu_parity.py::test_assembly_forms_give_the_same_system failed once in ~8 suite runs, round 5):
repeats build + solve on the test's scene under the default and the alternative assembly forms and reports every solve whose EO slice
differs from the first one's.   python scripts/eo_flake_probe.py [reps=300]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bundle_adjustment_amd import engine, scene
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
fp = scene.make_scene(12, 150, 90, dist=scene.DIST_FULL, weights="block", n_control=5, control_dense=True)
s2 = fp.sigma2apriori
ref = None
bad = 0
for rep in range(reps):
    form = (None, "t_vector", "no_fork", "materialise")[rep % 4]
    if form: os.environ["JAICOV_ASSEMBLY_FORM"] = form
    else: os.environ.pop("JAICOV_ASSEMBLY_FORM", None)
    eng = engine.Engine(fp)
    eng.set_parameters(fp.values)
    eng.build(s2, 0.5)
    dx = eng.solve(False)
    e0 = eng.reduced_order()
    eng.close()
    if ref is None:
        ref = dx
    d = np.abs(dx - ref)
    if d.max() > 1e-9 * np.abs(ref).max():
        bad += 1
        print(f"rep {rep} form {form}: {int((d > 1e-9 * np.abs(ref).max()).sum())} entries differ; first at {int(np.argmax(d > 1e-9 * np.abs(ref).max()))} (reduced order {e0}); zeros in the EO slice: {int((dx[e0:] == 0).sum())} of {dx.size - e0}", flush=True)
print(f"{reps} engines, {bad} with a deviating step", flush=True)
