"""Arrival-order (deterministic = off) vs deterministic (the default since round 4) assembly: ms per LM pass and per assembly stage at a config (GPU box).  python scripts/det_ab.py [cfg4] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bundle_adjustment_amd import engine, scene
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
fp = scene.config(cfg)
s2 = fp.sigma2apriori
out = {}
for det in (0, 1, 0, 1):
    eng = engine.Engine(fp, deterministic=bool(det))
    eng.set_parameters(fp.values)
    for _ in range(2):
        eng.build(s2, 0.0); eng.solve(False)
    asm = tot = 0.0
    t = time.perf_counter()
    for _ in range(steps):
        eng.build(s2, 0.0); eng.solve(False)
        tm = eng.timings(); asm += tm["assembly"]; tot += tm["total"]
    wall = 1e3 * (time.perf_counter() - t) / steps
    N1, n1 = None, None
    if det:
        eng.build(s2, 0.0); a = eng.get_normal(); eng.build(s2, 0.0); b = eng.get_normal()
        same = bool(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]))
    else:
        same = None
    print(f"{cfg} deterministic={det}: {wall:.3f} ms per pass (wall), assembly stage {asm / steps:.3f} ms, device total {tot / steps:.3f} ms, identical bits on rebuild: {same}", flush=True)
    eng.close()
