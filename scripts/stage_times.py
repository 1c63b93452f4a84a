"""Stage timings of one LM pass on the GPU for a named config (exploration helper, not the bench)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bundle_adjustment_amd import engine, scene

names = sys.argv[1:] or ["cfg2", "cfg3"]
for name in names:
    t = time.time(); fp = scene.config(name); tg = time.time() - t
    t = time.time(); eng = engine.Engine(fp); tc = time.time() - t
    eng.set_parameters(fp.values)
    s2 = fp.sigma2apriori
    print(f"{name}: U={fp.n_unknowns} n_ip={fp.n_image_points} gen={tg:.1f}s create={tc:.2f}s", flush=True)
    for it in range(4):
        t = time.time()
        eng.build(s2, 0.0); dx = eng.solve(False); mx = eng.update(dx)
        wall = time.time() - t
        tm = eng.timings()
        print(f"  it{it} wall={wall*1e3:.1f}ms max|dx|={mx:.3e} " + " ".join(f"{k}={v:.2f}" for k, v in tm.items()), flush=True)
    t = time.time()
    eng.prepare_inverse(True); eng.build(s2, 0.0); dx = eng.solve(True); om = eng.omega(s2, dx)
    wall = time.time() - t
    tm = eng.timings()
    print(f"  final(invert) wall={wall*1e3:.1f}ms omega={om:.4e} s0ratio={om/fp.degree_of_freedom/s2:.3f} " + " ".join(f"{k}={v:.2f}" for k, v in tm.items()), flush=True)
    eng.close()
