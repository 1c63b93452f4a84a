"""Does the history of a process matter to the dataflow factorisation?  N small engines are created, used and closed (optionally with
failing creations in between), then an ordinary problem of 94 block columns is factorised 20 times: abandoned factorisations and ms.
usage: engine_count_probe.py N [bad]   (one process per setting; GPU box)"""
import os, sys, time, dataclasses
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bundle_adjustment_amd import engine, scene
n = int(sys.argv[1]); bad = len(sys.argv) > 2 and sys.argv[2] == "bad"
small = scene.make_scene(6, 40, 24, dist=scene.DIST_FULL, weights="block", n_control=4)
for i in range(n):
    if bad and i % 2:
        try:
            engine.Engine(dataclasses.replace(small, datum_flags=small.datum_flags | 1))
        except engine.EngineError:
            pass
        continue
    e = engine.Engine(small); e.set_parameters(small.values); e.build(small.sigma2apriori, 0.0); e.solve(False); e.close()
fp = scene.make_scene(140, 4000, 2000, dist=scene.DIST_RADIAL, weights="2x2", n_control=6)
eng = engine.Engine(fp, ordinary_group_elimination=1); eng.set_parameters(fp.values)
t = time.perf_counter()
for _ in range(20):
    eng.build(fp.sigma2apriori, 0.0); eng.solve(False)
ms = 1e3 * (time.perf_counter() - t) / 20
st = eng.kernel_stats()
print(f"{n} engines before{' (half of them failing creations)' if bad else ''}: {ms:.2f} ms per pass, abandoned factorisations {st['flow_retries']}", flush=True)
eng.close()
