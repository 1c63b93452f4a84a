#!/usr/bin/env python
"""WHAT in the oracle's (= the reference algorithm's) assembly is 2e-12 .. 2e-11 away from the exactly assembled normal equations?
CPU only (oracle + ground truth, no device):  python scripts/exactN_attribution.py        (config-3 size with dense dispersions, ~4 min)

Assembles N at the converged point of tests/golden/cfg3b three times with the oracle's OWN fp64 accumulation (oracle_block_fair) and
different weights, and compares the probe N.v with the extended-precision truth (tests/golden/cfg3b/cfg3b_exactN.npz):
    weights = the reference's dpptrf + dpptri of D / sigma0^2 (DOPG:82-86)      N.v error 2.0e-12
    weights = the extended-precision inverse rounded to fp64                    N.v error 4.6e-15
    weights = LAPACK's LU inverse of D, scaled                                  N.v error 9.0e-15
so the accumulation is innocent and the Cholesky-based fp64 inverse of an ill-conditioned dispersion (cond 1e6 .. 2e7) is what the
covariances inherit.  The device's inverses carry one Newton-Schulz step with an error-free residual (batchinv.hip): 2e-16."""
import concurrent.futures as cf
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_cfg4_golden as g  # noqa: E402

orc = g.orc
fp = g.scene.config("cfg3_block")
o = orc.Oracle(fp)
s2, U = fp.sigma2apriori, fp.n_unknowns
z = np.load(os.path.join(ROOT, "tests", "golden", "cfg3b", "cfg3b_converged.npz"))
t = np.load(os.path.join(ROOT, "tests", "golden", "cfg3b", "cfg3b_exactN.npz"))
values, probe, truth = z["values"], g.probe_vector(U), t["Nv_exact"]


def err(N):
    return float(np.abs(g.packed_matvec(N, probe) - truth).max() / np.abs(truth).max())


with cf.ThreadPoolExecutor(6) as ex:
    w_ref = list(ex.map(lambda b: o.block_weight(s2, b), range(fp.n_image_blocks)))
    w_ext = list(ex.map(lambda b: o.exact_block_weight(s2, b)[0], range(fp.n_image_blocks)))
w_lu = []
for b in range(fp.n_image_blocks):
    m = w_ref[b].shape[0]
    D = fp.blk_disp[fp.blk_disp_offset[b]:fp.blk_disp_offset[b] + m * m].reshape(m, m)
    w_lu.append(np.ascontiguousarray(s2 * np.linalg.inv(D)))
for name, w in (("dpptrf + dpptri of D / sigma0^2 (the reference)", w_ref), ("extended-precision inverse, rounded to fp64", w_ext),
                ("LAPACK LU inverse of D, scaled", w_lu)):
    N, _ = g.assemble(o, fp, values, s2, w, [])
    print(f"weights = {name}: N.v error {err(N):.2e};  weights of block 0 against the extended ones {np.abs(w[0] - w_ext[0]).max() / np.abs(w_ext[0]).max():.2e}")
