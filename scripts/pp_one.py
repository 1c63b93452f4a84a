"""three builds + solves at config 4 with the gather's environment switches as given (for rocprofv3 --pmc)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundle_adjustment_amd import engine, scene
fp = scene.config("cfg4"); eng = engine.Engine(fp); eng.set_parameters(fp.values)
for i in range(3):
    eng.build(fp.sigma2apriori, 0.0)
eng.close()
