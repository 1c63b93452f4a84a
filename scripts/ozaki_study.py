"""Numerical half of the exploration "can the trailing update beat the fp64 pipe?" (the hardware half is
scripts/micro/ozaki_probe.hip).  CPU only, numpy:   python scripts/ozaki_study.py [n=1536]

An SPD matrix with the conditioning of the config-3b system (Jacobi-scaled, cond ~ 4e8) is factored by a left-looking blocked Cholesky
(block 128, as cholflow.hip) whose update products  C -= L[i][k] L[j][k]'  are formed (a) in fp64, (b) by the error-free sliced scheme the int8
matrix cores would run: every 128 x 64 k-block of a finished L tile is cut ONCE into S slices of 7 bits (signed, relative to the power of
two above the block row's largest entry), a product is the sum over the slice pairs p + q <= S - 1 of exact integer products (int32 on the
device; float64 matmuls of integer-valued arrays here, exact below 2^53), scaled by 2^(-7 (p + q + 2)) and the two row scales and added in
fp64.  Reported: the factor's backward error |M - L L'| / (|L||L'|) and the error of the solve (one right-hand side, exact solution by
extended-precision refinement) without and with refinement steps on the fp64 residual -- which is what the product path would use
(refine.hip): slices needed for the fp64 factor's accuracy, and slices needed when the refinement that already runs makes up the rest."""
import sys
import numpy as np

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
NB, KB = 128, 64
rng = np.random.default_rng(5)


def spd(n, cond):
    q, _ = np.linalg.qr(rng.normal(size=(n, n)))
    ev = np.logspace(0, -np.log10(cond), n)
    M = (q * ev) @ q.T
    d = 1.0 / np.sqrt(np.diag(M))
    M = M * d[:, None] * d[None, :]                     # Jacobi scaling (NES.applyPrecondition)
    return 0.5 * (M + M.T)


def slice_rows(T, S):
    """T (rows x 64) -> S integer-valued float arrays + per-row scale 2^e: T ~ scale * sum_p sl[p] 2^(-7 (p + 1)), |sl[p]| <= 64"""
    amax = np.abs(T).max(axis=1)
    e = np.where(amax > 0, np.ceil(np.log2(np.maximum(amax, 1e-300))) + 1, 0.0)
    sc = np.exp2(e)
    r = T / sc[:, None]                                 # |r| < 1/2
    out = []
    for p in range(S):
        s = np.rint(r * 128.0)                          # 7 bits + sign
        out.append(s)
        r = r * 128.0 - s
    return out, sc


def sliced_product(A_sl, a_sc, B_sl, b_sc, S):
    C = 0.0
    for g in range(S):                                  # scale groups p + q = g
        acc = 0.0
        for p in range(g + 1):
            acc = acc + A_sl[p] @ B_sl[g - p].T         # exact: integers, |sum| < 64 * 64 * 64 * S
        C = C + acc * 2.0 ** (-7 * (g + 2))
    return C * a_sc[:, None] * b_sc[None, :]


def cholesky(M, S):
    n = M.shape[0]
    nb = n // NB
    L = np.zeros_like(M)
    slices = {}                                         # (block row, k-block of 64) -> (slices, scales), cut once per finished tile
    for j in range(nb):
        for i in range(j, nb):
            C = M[i * NB:(i + 1) * NB, j * NB:(j + 1) * NB].copy()
            for k in range(j):
                if S == 0:
                    C -= L[i * NB:(i + 1) * NB, k * NB:(k + 1) * NB] @ L[j * NB:(j + 1) * NB, k * NB:(k + 1) * NB].T
                else:
                    for h in range(NB // KB):
                        a, asc = slices[(i, k, h)]
                        b, bsc = slices[(j, k, h)]
                        C -= sliced_product(a, asc, b, bsc, S)
            if i == j:
                L[j * NB:(j + 1) * NB, j * NB:(j + 1) * NB] = np.linalg.cholesky(C)
            else:
                Ljj = L[j * NB:(j + 1) * NB, j * NB:(j + 1) * NB]
                L[i * NB:(i + 1) * NB, j * NB:(j + 1) * NB] = np.linalg.solve(Ljj, C.T).T
            if S:
                for h in range(NB // KB):
                    slices[(i, j, h)] = slice_rows(L[i * NB:(i + 1) * NB, j * NB + h * KB:j * NB + (h + 1) * KB], S)
    return L


def solve(L, b):
    return np.linalg.solve(L.T, np.linalg.solve(L, b))


M = spd(n, 4e8)
xt = rng.normal(size=n)
Ml = M.astype(np.longdouble)
b = np.asarray(Ml @ xt.astype(np.longdouble), dtype=np.float64)
# exact solution of the fp64 system M x = b by extended-precision refinement on the fp64 factor
Lref = np.linalg.cholesky(M)
x = solve(Lref, b)
for _ in range(6):
    r = np.asarray(b.astype(np.longdouble) - Ml @ x.astype(np.longdouble), dtype=np.float64)
    x = x + solve(Lref, r)
xex = x
print(f"order {n}, cond(M) = {np.linalg.cond(M):.2e}; backward error = max |M - L L'| / (|L| |L'|); solve error = max |x - x_exact| / max |x_exact|, "
      "after 0 / 1 / 2 / 3 steps of refinement with fp64 residuals")
for S in (0, 9, 8, 7, 6, 5, 4):
    L = cholesky(M, S)
    bw = (np.abs(M - L @ L.T) / (np.abs(L) @ np.abs(L).T)).max()
    errs = []
    x = solve(L, b)
    for it in range(4):
        errs.append(np.abs(x - xex).max() / np.abs(xex).max())
        x = x + solve(L, b - M @ x)
    pairs = S * (S + 1) // 2
    print(f"  {'fp64 products' if S == 0 else f'{S} slices ({pairs:2d} pairs)':22s}: backward error {bw:.1e}; solve error " + " / ".join(f"{e:.1e}" for e in errs))
