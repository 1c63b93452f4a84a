"""One pass of the densified J'WJ contraction (assembly_mode = 1) at config 4, for a rocprofv3 --kernel-trace --stats summary of
the batched gemm_f64_kernel<0,1> / <1,1> launches that bench.py's `jtwj_dense_mode` figure is made of."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundle_adjustment_amd import engine, scene
fp = scene.config("cfg4")
de = engine.Engine(fp, assembly_mode=1)
de.set_parameters(fp.values)
for _ in range(3):
    de.accumulate(fp.sigma2apriori)
de.close()
