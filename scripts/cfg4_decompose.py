"""Where does the difference between the GPU's step and the oracle's at config 4 (2e-8 .. 2e-7 relative, cond(V N V) ~ 1e9) come
from: the assembly's rounding (both N are roundings of the same exact sums, they differ by ~1e-11) or the solvers?  Takes the GPU's
own N, n to the host, solves THAT system with the oracle's packed Bunch-Kaufman (dspsv) and with extended-precision refinement
(exact solution), and compares everything.  ~10 min of one host core.  python scripts/cfg4_decompose.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle as orc
from bundle_adjustment_amd import engine, scene

G = os.path.join(ROOT, "tests", "golden", "cfg4")
z = np.load(os.path.join(G, "cfg4_oracle.npz"))
fp = scene.config("cfg4"); U = fp.n_unknowns; s2 = fp.sigma2apriori
rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
eng = engine.Engine(fp); eng.set_parameters(fp.values)
eng.build(s2, 0.0); dx_red = eng.solve(False)
eng.prepare_inverse(engine.INVERT_FULL); eng.build(s2, 0.0)
N, n = eng.get_normal(); dx_full = eng.solve(False)
eng.close()
print(f"GPU reduced vs GPU full-order: {rel(dx_red, dx_full):.3e}; vs oracle fixture: reduced {rel(dx_red, z['dx1']):.3e} full {rel(dx_full, z['dx1']):.3e}", flush=True)
o = orc.Oracle(fp); L = orc.lib()
V = o.finalize(fp.values, N, n, 0.0, False)
o.precondition(V, N, n)
A0 = N.copy(); b = n.copy()
ipiv = np.zeros(U, np.int32)
t = time.time(); info = L.oracle_dsptrf(U, orc._p(N), ipiv.ctypes.data_as(orc._pi)); assert info == 0
print(f"dsptrf on the GPU's N: {time.time() - t:.0f} s", flush=True)
y = b.copy(); L.oracle_dsptrs(U, orc._p(N), ipiv.ctypes.data_as(orc._pi), orc._p(y))
dx_bk = V * y
r = np.zeros(U)
for it in range(4):
    L.oracle_residual_ld(U, orc._p(A0), orc._p(y), orc._p(b), orc._p(r))
    L.oracle_dsptrs(U, orc._p(N), ipiv.ctypes.data_as(orc._pi), orc._p(r))
    y = y + r
    print(f"  refinement {it}: correction {np.abs(r).max() / np.abs(y).max():.3e}", flush=True)
dx_exact = V * y
out = {"gpu_full_vs_exact_solution_of_gpu_N": rel(dx_full, dx_exact), "gpu_reduced_vs_exact_solution_of_gpu_N": rel(dx_red, dx_exact),
       "oracle_dspsv_on_gpu_N_vs_exact": rel(dx_bk, dx_exact), "exact_of_gpu_N_vs_oracle_fixture_dx1": rel(dx_exact, z["dx1"]),
       "oracle_dspsv_on_gpu_N_vs_oracle_fixture_dx1": rel(dx_bk, z["dx1"])}
print(json.dumps(out, indent=1), flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "cfg4_decompose.json"), "w"), indent=1)
