"""Engine creation at config 4 (4 GB of dense dispersions -> weights) and the whole adjustment behind it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundle_adjustment_amd import engine, scene
fp = scene.config(sys.argv[1] if len(sys.argv) > 1 else "cfg4")
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    t = time.perf_counter(); eng = engine.Engine(fp); t1 = time.perf_counter() - t
    ct = eng.create_timings()
    t = time.perf_counter(); v, r = eng.estimate(invert=engine.INVERT_FULL); t2 = time.perf_counter() - t
    ks = eng.kernel_stats()
    print(f"create {t1:.3f} s {ct}, estimate to termination (FULL) {t2:.3f} s, iterations {r.iterations}, seconds_total {r.seconds_total:.3f}, "
          f"last pass {r.seconds_last_pass:.3f} s, flow retries {ks['flow_retries']}, slow-path flag hits {ks['flow_stale_events']}", flush=True)
    eng.close()
