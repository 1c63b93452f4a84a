import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from bundle_adjustment_amd import engine, scene
fp = scene.config("cfg4")
t = time.perf_counter(); e1 = engine.Engine(fp); print("e1", round(time.perf_counter() - t, 3), flush=True)
e1.set_parameters(fp.values); e1.build(fp.sigma2apriori, 0.0); e1.solve(False)
for i in range(3):
    t = time.perf_counter(); e2 = engine.Engine(fp); w = time.perf_counter() - t
    print("e2 while e1 lives", round(w, 3), {k: round(v, 1) for k, v in e2.create_timings().items()}, flush=True)
    t = time.perf_counter(); e2.close(); print("   close", round(time.perf_counter() - t, 3), flush=True)
e1.close()
