import ctypes as C, sys, os
sys.path.insert(0, '/root/repo')
import numpy as np
from bundle_adjustment_amd import engine
L = engine.load_library()
out = (C.c_int * 8)()
print("rc", L.jaicov_debug_flow_residency(out), list(out))
