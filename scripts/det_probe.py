import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if os.environ.get('PROBE_TORCH'):
    import torch; torch.cuda.is_available()
from bundle_adjustment_amd import engine, scene
fp = scene.make_scene(12, 150, 80, dist=scene.DIST_FULL, weights="block", n_control=6, control_dense=True)
s2 = fp.sigma2apriori; U = fp.n_unknowns
print("U", U, "e0", int(fp.eo_col.min()), "io", fp.io_col.ravel(), "dist", fp.dist_col, "ctrl slots", fp.dg_slot[:6])
res = []
def rc(idx):
    c = int((np.sqrt(8 * idx + 1) - 1) / 2)
    while (c + 1) * (c + 2) // 2 <= idx: c += 1
    while c * (c + 1) // 2 > idx: c -= 1
    return idx - c * (c + 1) // 2, c
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    eng = engine.Engine(fp, deterministic=True); eng.set_parameters(fp.values)
    eng.build(s2, 0.0); N, n = eng.get_normal()
    eng.build(s2, 0.0); N2, n2 = eng.get_normal()
    if not np.array_equal(N, N2): print('same engine, second build differs at', [rc(int(i)) for i in np.flatnonzero(N != N2)[:8]])
    dx = eng.solve(False)
    eng.prepare_inverse(engine.INVERT_FULL); eng.build(s2, 0.0); Nf, nf = eng.get_normal()
    res.append((N, n, Nf, nf)); eng.close()
for k, name in enumerate(("N reduced", "n reduced", "N full", "n full")):
    for r in res[1:]:
        d = np.flatnonzero(r[k] != res[0][k])
        if d.size:
            print(name, "differs at", [(rc(int(i)) if k in (0, 2) else int(i)) for i in d[:12]], "values", res[0][k][d[:4]], r[k][d[:4]])
