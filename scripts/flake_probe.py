import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle")); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
torch.cuda.is_available()
import oracle
oracle.build()
from bundle_adjustment_amd import engine, scene
from bundle_adjustment_amd.problem import packed_to_full
fp = scene.config("tiny_block")
o = oracle.Oracle(fp); s2 = fp.sigma2apriori
dxo, Qo, _, _ = o.step(fp.values, s2, 0.0, True)
U = fp.n_unknowns; d = fp.rank_defect
Qref = packed_to_full(Qo, U); sd = np.sqrt(np.abs(np.diag(Qref)))
worst = [0, 0, 0, 0]
for it in range(300):
    eng = engine.Engine(fp); eng.set_parameters(fp.values); eng.prepare_inverse(True); eng.build(s2, 0.0)
    dx = eng.solve(True)
    Q = packed_to_full(eng.get_cofactor(), U)
    a = np.abs(dx - dxo).max() / np.abs(dxo).max()
    b = (np.abs(Q - Qref) / np.outer(sd, sd)).max()
    c = (np.abs(np.diag(Q) - np.diag(Qref)) / np.abs(np.diag(Qref))).max()
    e = np.abs(Q - Qref).max() / np.abs(Qref).max()
    worst = [max(worst[0], a), max(worst[1], b), max(worst[2], c), max(worst[3], e)]
    if not np.all(np.isfinite(dx)) or a > 1e-9 or b > 1e-9 or c > 1e-9 or e > 1e-8:
        print("iter", it, "dx", a, "Qscaled", b, "diag", c, "Qabs", e, flush=True)
    eng.close()
print("worst", worst)
