"""Factorisation time of config 4 against the outer panel width (JAICOV_NBO), one process."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundle_adjustment_amd import engine, scene

fp = scene.config(sys.argv[1] if len(sys.argv) > 1 else "cfg4")
eng = engine.Engine(fp)
eng.set_parameters(fp.values)
s2 = fp.sigma2apriori
for it in range(3):
    eng.build(s2, 0.0); dx = eng.solve(False); eng.update(dx)
for nbo in (256, 384, 512, 640, 768, 1024, 512):
    os.environ["JAICOV_NBO"] = str(nbo)
    best = None
    for rep in range(3):
        eng.build(s2, 0.0); dx = eng.solve(False)
        tm = eng.timings()
        best = tm if best is None or tm["factor"] < best["factor"] else best
    print(f"nbo={nbo} factor={best['factor']:.2f} ms solve={best['solve']:.2f}", flush=True)
eng.close()
