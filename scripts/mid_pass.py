"""A mid-size adjustment (60 images x 600 points, 2 x 2 weights: reduced order ~1 810 = 15 block columns): ms per LM pass and its stages.
    python scripts/mid_pass.py [n_images=60] [n_points=600] [obs=300]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bundle_adjustment_amd import engine, scene
ni, npt, ob = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 60), (2, 600), (3, 300)))
fp = scene.make_scene(ni, npt, ob, dist=scene.DIST_FULL, weights="2x2", n_control=6)
eng = engine.Engine(fp)
eng.set_parameters(fp.values)
s2 = fp.sigma2apriori
for _ in range(5):
    eng.build(s2, 0.0); eng.solve(False)
t = time.perf_counter(); st = np.zeros(8)
n = 100
for _ in range(n):
    eng.build(s2, 0.0); eng.solve(False)
    st += np.array(list(eng.timings().values()))
wall = (time.perf_counter() - t) / n * 1e3
print(f"U={fp.n_unknowns} reduced order {eng.reduced_order()}: {wall:.3f} ms per pass (wall); stages rows, assembly, finalize, factor, solve, inverse, omega, total (device, ms): {np.round(st / n, 3)}")
eng.close()
