"""Which stage of the device's SPD path loses digits against the reference's packed Bunch-Kaufman?

CPU experiment (numpy, no GPU): the oracle's Jacobi-scaled normal matrix M = V N V of a config-3-sized scene with config 4's
dense dispersions (U = 3 614, cond ~ 1e9).  Ground truth by iterative refinement with long-double residuals.  Variants:

  factor     A  blocked Cholesky, 128 blocks, off-diagonal tiles finished by a PRODUCT with the explicit inverse of the diagonal
                block (what cholflow.hip does)
             B  the same with a triangular solve (TRSM)
             C  LAPACK dpotrf
  solve      forward + back substitution on each factor            -> forward error of x
  inverse    1  W = L^-1 by recursive products of block inverses (dense.hip trtri), Q = W'W
             2  W = L^-1 by substitution (column-wise triangular solves, dtrtri), Q = W'W
             3  the oracle's dsptrf + dsptri (the reference's algorithm)
  refine     one step on x with the residual in fp64 / in double-double (emulated with long double)
             one Newton-Schulz step on Q:  Q <- Q + Q (I - M Q)

usage: python scripts/stability_probe.py [mid|cfg3_block]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import scipy.linalg as sl
import oracle as orc
from bundle_adjustment_amd import scene

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3_block"
if name == "mid":
    fp = scene.make_scene(12, 150, 90, dist=scene.DIST_FULL, weights="block", n_control=5, control_dense=True)
else:
    fp = scene.make_scene(100, 1000, 400, dist=scene.DIST_FULL, weights="block", n_control=15, control_dense=True)
U, s2 = fp.n_unknowns, fp.sigma2apriori
o = orc.Oracle(fp); Lb = orc.lib()
t = time.time()
Np = np.zeros(fp.packed_length); n = np.zeros(U)          # image groups through the fair two-product form (make_cfg4_golden.py)
for blk in range(fp.n_image_blocks):
    assert o.block_fair(fp.values, s2, blk, o.block_weight(s2, blk), Np, n) == 0
Ns, ns = o.accumulate(fp.values, s2, 0, 0, shared=True)
Np += Ns; n += ns; del Ns
V = o.finalize(fp.values, Np, n, 0.0, False)
print(f"U = {U}, oracle assembly {time.time() - t:.1f} s", flush=True)
o.precondition(V, Np, n)
M = np.zeros((U, U)); iu = np.triu_indices(U)
M.T[iu[1], iu[0]] = 0
M[iu] = Np[(iu[0] + iu[1] * (iu[1] + 1) // 2)]
M = np.triu(M) + np.triu(M, 1).T
b = n.copy()
ev = np.linalg.eigvalsh(M)
print(f"cond(M) = {ev[-1] / ev[0]:.3e}", flush=True)
rel = lambda a, r: float(np.abs(a - r).max() / np.abs(r).max())
NB = 128
Ml = M.astype(np.longdouble)


def refine_ld(x, solve, rhs, iters=5):
    for _ in range(iters):
        r = (rhs.astype(np.longdouble) - Ml @ x.astype(np.longdouble)).astype(np.float64)
        dx = solve(r)
        x = x + dx
        if np.abs(dx).max() < 1e-17 * np.abs(x).max():
            break
    return x


def chol_blocked(A, explicit):
    n_ = A.shape[0]; L = np.tril(A.copy())
    for j in range(0, n_, NB):
        je = min(j + NB, n_)
        L[j:je, j:je] -= L[j:je, :j] @ L[j:je, :j].T
        Ljj = np.linalg.cholesky(np.tril(L[j:je, j:je]) + np.tril(L[j:je, j:je], -1).T)
        L[j:je, j:je] = Ljj
        if je < n_:
            T = L[je:, j:je] - L[je:, :j] @ L[j:je, :j].T
            if explicit:
                inv = sl.solve_triangular(Ljj, np.eye(je - j), lower=True)
                L[je:, j:je] = T @ inv.T
            else:
                L[je:, j:je] = sl.solve_triangular(Ljj, T.T, lower=True).T
    return L


def trtri_recursive(L):
    """inverse of lower-triangular L by the block recursion inv([[A,0],[C,B]]) = [[A^-1,0],[-B^-1 C A^-1, B^-1]], leaves = 128"""
    n_ = L.shape[0]
    if n_ <= NB:
        return sl.solve_triangular(L, np.eye(n_), lower=True)
    h = (n_ // NB + 1) // 2 * NB
    Ai = trtri_recursive(L[:h, :h]); Bi = trtri_recursive(L[h:, h:])
    W = np.zeros_like(L)
    W[:h, :h] = Ai; W[h:, h:] = Bi; W[h:, :h] = -Bi @ (L[h:, :h] @ Ai)
    return W


F = {"A explicit-inverse tiles": chol_blocked(M, True), "B trsm tiles": chol_blocked(M, False), "C dpotrf": np.linalg.cholesky(M)}
solveC = lambda r: sl.cho_solve((F["C dpotrf"], True), r)
x_true = refine_ld(solveC(b), solveC, b)
print("\n-- solve: forward error of x against the exact solution")
for k, L in F.items():
    x = sl.solve_triangular(L, sl.solve_triangular(L, b, lower=True), lower=True, trans="T")
    r64 = b - M @ x
    x1 = x + sl.solve_triangular(L, sl.solve_triangular(L, r64, lower=True), lower=True, trans="T")
    rdd = (b.astype(np.longdouble) - Ml @ x.astype(np.longdouble)).astype(np.float64)
    x2 = x + sl.solve_triangular(L, sl.solve_triangular(L, rdd, lower=True), lower=True, trans="T")
    print(f"  {k:28s} {rel(x, x_true):.2e}   +1 step fp64 residual {rel(x1, x_true):.2e}   +1 step extended residual {rel(x2, x_true):.2e}")
# the reference's algorithm
Nf = Np.copy(); ipiv = np.zeros(U, np.int32)
assert Lb.oracle_dsptrf(U, orc._p(Nf), ipiv.ctypes.data_as(orc._pi)) == 0
xb = b.copy(); Lb.oracle_dsptrs(U, orc._p(Nf), ipiv.ctypes.data_as(orc._pi), orc._p(xb))
print(f"  {'oracle dspsv (reference)':28s} {rel(xb, x_true):.2e}")

print("\n-- inverse: error against exact columns (64 columns sampled), relative to max |Q|, and diag relative")
rng = np.random.default_rng(1); cols = np.sort(rng.choice(U, 64, replace=False))
Qt = np.zeros((U, cols.size))
for a, c in enumerate(cols):
    e = np.zeros(U); e[c] = 1.0
    Qt[:, a] = refine_ld(solveC(e), solveC, e)
qmax = np.abs(Qt).max()
dtrue = np.array([Qt[c, a] for a, c in enumerate(cols)])


def report(label, Q):
    ec = np.abs(Q[:, cols] - Qt).max() / qmax
    ed = np.abs(np.array([Q[c, c] for c in cols]) / dtrue - 1).max()
    sd = np.sqrt(np.abs(np.diag(Q)))
    ecorr = (np.abs(Q[:, cols] - Qt) / np.outer(sd, sd[cols])).max()
    print(f"  {label:58s} columns {ec:.2e}   diag {ed:.2e}   correlation-scaled {ecorr:.2e}", flush=True)
    return Q


for k, L in F.items():
    W1 = trtri_recursive(L); Q1 = report(f"{k} | W recursive block products | W'W", W1.T @ W1)
    W2 = sl.solve_triangular(L, np.eye(U), lower=True); report(f"{k} | W by substitution | W'W", W2.T @ W2)
    if k.startswith("A"):
        QA = Q1
Qp = Nf.copy(); work = np.zeros(U)
assert Lb.oracle_dsptri(U, orc._p(Qp), ipiv.ctypes.data_as(orc._pi), orc._p(work)) == 0
Qb = np.zeros((U, U)); Qb[iu] = Qp[(iu[0] + iu[1] * (iu[1] + 1) // 2)]; Qb = np.triu(Qb) + np.triu(Qb, 1).T
report("oracle dsptrf + dsptri (reference)", Qb)
# Newton-Schulz on the device-like inverse
R = np.eye(U) - M @ QA
report("A | recursive | + Newton-Schulz step, fp64 residual", QA + QA @ R)
Rl = (np.eye(U, dtype=np.longdouble) - Ml @ QA.astype(np.longdouble)).astype(np.float64)
report("A | recursive | + Newton-Schulz step, extended residual", QA + QA @ Rl)
