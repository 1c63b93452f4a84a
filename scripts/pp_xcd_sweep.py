"""Point x point gather: chunk width x XCD-partitioned mapping (JAICOV_PP_CW, JAICOV_PP_XCD are read at engine creation).
One child process per variant: assembly stage time of 10 builds at config 4.  usage: pp_xcd_sweep.py [cw:xcd ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from bundle_adjustment_amd import engine, scene
fp = scene.config("cfg4"); eng = engine.Engine(fp); eng.set_parameters(fp.values); s2 = fp.sigma2apriori
t = []
for i in range(9):
    eng.build(s2, 0.0)
    dx = eng.solve(False)            # the stage times are read out at the end of a solve
    if i >= 3: t.append(eng.timings()["assembly"])
print("assembly %%.3f ms (min %%.3f)  |dx|max %%.9e" %% (np.mean(t), np.min(t), np.abs(dx).max()), flush=True)
eng.close()
''' % ROOT
variants = sys.argv[1:] or ["1664:0", "1664:1", "960:0", "960:1", "640:1", "1280:1"]
for v in variants:
    cw, x = v.split(":")
    env = dict(os.environ, JAICOV_PP_CW=cw, JAICOV_PP_XCD=x)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    print(f"cw={cw:5s} xcd_map={x}: {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)
