"""Assembly stage (deterministic default) at config 4 and on the strips scene, one engine each: python scripts/asm_two_scenes.py  (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundle_adjustment_amd import engine, scene
for cfg in ("cfg4", "cfg4_local"):
    fp = scene.config(cfg)
    eng = engine.Engine(fp); eng.set_parameters(fp.values)
    best = 1e9; tot = 0.0
    for i in range(12):
        eng.build(fp.sigma2apriori, 0.0); eng.solve(False)
        if i >= 2:
            a = eng.timings()["assembly"]; best = min(best, a); tot += a / 10
    print(f"{cfg}: assembly stage mean {tot:.3f} ms, best {best:.3f} ms", flush=True)
    eng.close()
