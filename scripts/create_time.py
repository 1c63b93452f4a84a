import sys, time
sys.path.insert(0, "/root/repo")
from bundle_adjustment_amd import engine, scene
fp = scene.config("cfg4")
for i in range(2):
    t = time.perf_counter(); eng = engine.Engine(fp); t1 = time.perf_counter() - t
    t = time.perf_counter(); v, r = eng.estimate(invert=engine.INVERT_FULL); t2 = time.perf_counter() - t
    print(f"create {t1:.3f} s, estimate to termination (FULL) {t2:.3f} s, iterations {r.iterations}, seconds_total {r.seconds_total:.3f}", flush=True)
    eng.close()
