"""Achieved differences between the HIP paths and the CPU oracle's fixture at the headline size (tests/golden/cfg4).
Prints the numbers the tolerances of tests/test_gpu_cfg4_golden.py are set from."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bundle_adjustment_amd import engine, scene

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cfg4")
z = np.load(os.path.join(G, "cfg4_oracle.npz")); meta = json.load(open(os.path.join(G, "cfg4_oracle.json")))
fp = scene.config("cfg4"); U = fp.n_unknowns; s2 = fp.sigma2apriori
probe = np.random.Generator(np.random.Philox(meta["probe_seed"])).standard_normal(U)


def rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


def packed_matvec(ap, v):
    y = np.zeros(v.size); off = 0
    for r in range(v.size):
        row = ap[off:off + r + 1]; y[r] += row @ v[:r + 1]; y[:r] += row[:r] * v[r]; off += r + 1
    return y


def update(values, dx):
    cols = fp.slot_columns(); v = values.copy(); m = cols >= 0; v[m] += dx[cols[m]]; return v


t0 = time.time()
eng = engine.Engine(fp); print(f"engine created in {time.time() - t0:.1f} s", flush=True)
eng.set_parameters(fp.values)
eng.build(s2, 0.0); dx = eng.solve(False)
print("pass 1 default (EO pre-eliminated): dx rel", rel(dx, z["dx1"]), " scaled by step per class:",
      rel(dx[:15000], z["dx1"][:15000]), rel(dx[15014:], z["dx1"][15014:]), flush=True)
eng.prepare_inverse(engine.INVERT_FULL); eng.build(s2, 0.0)
N, n = eng.get_normal()
print("pass 1 full system: n rel", rel(n, z["n1"]), " N.v rel", rel(packed_matvec(N, probe), z["Nv1"]), flush=True)
del N
dxf = eng.solve(False)
print("pass 1 full-order solve: dx rel", rel(dxf, z["dx1"]), flush=True)
v1 = update(fp.values, z["dx1"])
for mode, name in ((engine.INVERT_FULL, "FULL"), (engine.INVERT_REDUCED, "REDUCED")):
    eng.set_parameters(v1); eng.prepare_inverse(mode); eng.build(s2, 0.0)
    dx2 = eng.solve(mode); om = eng.omega(s2, dx2)
    k = eng.cofactor_order()
    cols = z["sample_cols"]; sel = cols[cols < k]
    Qs = eng.get_cofactor_sub(sel.astype(np.int32))
    ref = z["Qsample"][np.ix_(cols < k, cols < k)]
    sd = np.sqrt(np.abs(np.diag(ref)))
    print(f"final pass {name}: order {k} dx2 rel {rel(dx2, z['dx2']):.3e} omega rel {abs(om - meta['omega']) / meta['omega']:.3e} "
          f"Qsample max|dQ|/sqrt(qii qjj) {np.abs((Qs - ref) / np.outer(sd, sd)).max():.3e} rel-to-max {rel(Qs, ref):.3e}", flush=True)
    Q = eng.get_cofactor()
    idx = np.arange(k, dtype=np.int64); dg = Q[idx * (idx + 3) // 2]
    print(f"   diag Q rel (elementwise max) {np.abs(dg / z['diagQ'][:k] - 1).max():.3e}", flush=True)
    if k == U:
        d = dg
        fro = float(np.sqrt(2.0 * np.dot(Q, Q) - np.dot(d, d)))
        print(f"   ||Q||_F rel {abs(fro - meta['qxx_frobenius']) / meta['qxx_frobenius']:.3e}  Q.v rel {rel(packed_matvec(Q, probe), z['Qv']):.3e}", flush=True)
    del Q
eng.close()
de = engine.Engine(fp, assembly_mode=1); de.set_parameters(fp.values); de.build(s2, 0.0); dxd = de.solve(False)
print("pass 1 dense-contraction mode: dx rel", rel(dxd, z["dx1"]), flush=True)
de.close()
