"""Link times of the substitution chains (JAICOV_CHAIN_TRACE, dense.hip): one refined solve at a config, the trace goes to stderr.
    JAICOV_CHAIN_TRACE=1 python scripts/chain_trace.py [config]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundle_adjustment_amd import engine, scene
name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
fp = scene.config(name)
eng = engine.Engine(fp)
eng.set_parameters(fp.values)
for _ in range(2):
    eng.build(fp.sigma2apriori, 0.0)
    eng.solve(engine.INVERT_NONE)
print(eng.timings())
eng.close()
