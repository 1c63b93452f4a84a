"""Final inverting passes only (for rocprofv3 --kernel-trace --stats): MatrixInversion.FULL as estimate() runs it
(JAICOV_INVERT_FULL_EXPANDED) and REDUCED, `reps` times each after one warm-up of each.  usage: final_pass.py [config] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundle_adjustment_amd import engine, scene

fp = scene.config(sys.argv[1] if len(sys.argv) > 1 else "cfg4")
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
eng = engine.Engine(fp)
eng.set_parameters(fp.values)
s2 = fp.sigma2apriori
for mode, name in ((engine.INVERT_FULL_EXPANDED, "FULL_EXPANDED"), (engine.INVERT_REDUCED, "REDUCED")):
    for r in range(reps + 1):
        t = time.perf_counter()
        eng.prepare_inverse(mode); eng.build(s2, 0.0); eng.solve(mode)
        if r:
            print(name, f"{1e3 * (time.perf_counter() - t):.2f} ms", {k: round(v, 2) for k, v in eng.timings().items()}, flush=True)
eng.close()
