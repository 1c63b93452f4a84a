"""Soak at an order between the configs (200 images x 2 500 points, dense dispersions: reduced order ~7 500 = 59 block columns, the chain
form with its third workgroup): python scripts/mid_soak.py [passes=2000]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bundle_adjustment_amd import engine, scene
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
fp = scene.make_scene(200, 2500, 400, dist=scene.DIST_FULL, weights="block", n_control=15, control_dense=True)
eng = engine.Engine(fp)
eng.set_parameters(fp.values)
s2 = fp.sigma2apriori
ref = None
for i in range(n):
    eng.build(s2, 0.0)
    dx = eng.solve(False)
    assert np.isfinite(dx).all(), i
    if ref is None:
        ref = dx
    elif i % 20 == 0:
        assert np.array_equal(dx, ref), i
    if i % 500 == 499:
        print(f"{i + 1} passes, reduced order {eng.reduced_order()}, {eng.timings()['total']:.2f} ms per pass", flush=True)
st = eng.kernel_stats()
eng.close()
assert st["flow_retries"] == 0, st
print("ok", st["flow_retries"], st["flow_stale_events"])
