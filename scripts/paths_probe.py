import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from bundle_adjustment_amd import engine, scene
from test_gpu_fullsize import _adjust
fp = scene.config("cfg4")
a = engine.Engine(fp)
P3, I6 = 3 * fp.n_points, 6 * fp.n_images
res = []
for rep in range(3):
    for mode in (engine.INVERT_NONE, engine.INVERT_FULL):
        dx1, v = _adjust(a, fp, mode)
        res.append((mode, dx1, v))
a.close()
v_full = res[1][2]
den = np.maximum(np.abs(v_full), 1.0); den[:P3] = 2000.0
eo = den[-I6:].reshape(-1, 6); eo[:, :3] = 2000.0
for mode, dx1, v in res:
    rel = np.abs(v - v_full) / den
    print(mode, "first step diff", np.abs(dx1 - res[1][1]).max() / np.abs(res[1][1]).max(), "converged rel", rel[:P3].max(), rel[P3:-I6].max(), rel[-I6:].max())
