"""Dense-contraction mode (assembly_mode=1): GEMM time per pass against the number of images per batched launch.
Each batch size runs in its own process (the engine reads JAICOV_DM_BATCH at create)."""
import json, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
CHILD = r'''
import importlib, json, sys, time, torch
sys.path.insert(0, %r)
from bundle_adjustment_amd import engine, scene
fp = scene.config("cfg4")
de = engine.Engine(fp, device=0, assembly_mode=1)
de.set_parameters(fp.values)
s2 = fp.sigma2apriori
de.accumulate(s2)
de.set_profiling(True); de.kernel_stats(reset=True)
torch.cuda.synchronize()
t = time.perf_counter(); de.accumulate(s2); torch.cuda.synchronize(); wall = 1e3 * (time.perf_counter() - t)
ks = de.kernel_stats()
print(json.dumps({"gemm_ms": ks["dense_gemm_ms"], "tflops": ks["dense_flops"] / ks["dense_gemm_ms"] / 1e9, "wall_ms": wall}))
''' % os.path.dirname(HERE)
for b in (sys.argv[1:] or ["16", "32", "64", "125", "128", "250", "500"]):
    env = dict(os.environ, JAICOV_DM_BATCH=b)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    print(b, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:], flush=True)
