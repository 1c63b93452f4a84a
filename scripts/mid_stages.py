import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from bundle_adjustment_amd import engine, scene
fp = scene.make_scene(200, 2500, 400, dist=scene.DIST_FULL, weights="block", n_control=15, control_dense=True)
eng = engine.Engine(fp); eng.set_parameters(fp.values); s2 = fp.sigma2apriori
for _ in range(5): eng.build(s2, 0.0); eng.solve(False)
st = np.zeros(8)
for _ in range(50):
    eng.build(s2, 0.0); eng.solve(False); st += np.array(list(eng.timings().values()))
print(dict(zip(eng.timings().keys(), np.round(st / 50, 3))))
