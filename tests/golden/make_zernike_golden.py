"""Generates tests/golden/jacobian_rows_zernike.json: independent golden vectors for the Zernike distortion rows.

Model function differentiated SYMBOLICALLY (sympy) and evaluated with mpmath at 50 digits, as make_jacobian_golden.py:
    Z(xs, ys) = len * sum_k c_k * rho^(n-2k) * G(phi),  rho^2 = (xs^2 + ys^2) / r0^2,  phi = atan2(ys, xs),
    G = cos(m phi) for m >= 0, sin(|m| phi) for m < 0          (ZernikeCoefficient.java:41-57,133-138)
    X model: dx = z * Z;  Y model: dy = z * Z;  Gradient model: (dx, dy) = z * grad Z
(ZernikeDistortionModelFactory.java:41-227).  Only polynomials of EVEN radial order n are used: for odd n the reference
truncates rho^(n-2k) to an even power (integer division, ZDF:107,176,178) while its chain-rule factors keep the odd
exponent, so no single function has the reference's value and the reference's derivative -- there the reference's
formulas are pinned by oracle-vs-kernel tests only.  Run once:  python tests/golden/make_zernike_golden.py
"""
import json
import math
import os

import mpmath as mp
import numpy as np
import sympy as sp

mp.mp.dps = 50
KINDS = {"ZX": 7, "ZY": 8, "ZZ": 9, "AI": 5}
SETS = {
    "zernike_x": [("ZX", 3), ("ZX", 4), ("ZX", 13)],
    "zernike_y": [("ZY", 5), ("ZY", 10), ("ZY", 12)],
    "zernike_gradient": [("ZZ", 3), ("ZZ", 5), ("ZZ", 11), ("ZZ", 24)],
    "zernike_mixed": [("AI", 1), ("ZX", 12), ("ZY", 14), ("ZZ", 4)],
    # round 5: the radial orders of the reference's third example (ExampleDistortionModel.java:95-99: single indices 4, 12, 24, 40, 60 =
    # Z_2^0 .. Z_10^0: six radial terms, binomials up to C(10, 5), rho^10) -- for the gradient model as the example uses them, and the two
    # highest for the X and Y models.  (Appended: the sets above keep their random draws, the file's earlier sets are unchanged.)
    "zernike_high_x": [("ZX", 40), ("ZX", 60)],
    "zernike_high_y": [("ZY", 40), ("ZY", 60)],
    "zernike_example_gradient": [("ZZ", 4), ("ZZ", 12), ("ZZ", 24), ("ZZ", 40), ("ZZ", 60)],
}


def zernike(order, u, v, r0):
    n = math.ceil((-3 + math.sqrt(9 + 8 * order)) / 2)
    m = 2 * order - n * (n + 2)
    assert n % 2 == 0, "even radial orders only (see module docstring)"
    half = (n - abs(m)) // 2
    rho2 = (u * u + v * v) / (r0 * r0)
    R = 0
    for k in range(half + 1):
        c = (-1) ** k * math.comb(n - k, k) * math.comb(n - 2 * k, half - k)
        R += c * rho2 ** sp.Rational(n - 2 * k, 2)
    length = sp.sqrt(sp.Integer((1 + (1 if m != 0 else 0)) * (n + 1)) / sp.pi)
    phi = sp.atan2(v, u)
    G = sp.cos(m * phi) if m >= 0 else sp.sin(-m * phi)
    return length * R * G


def model(dist):
    X, Y, Z, x0, y0, c, X0, Y0, Z0, om, ph, ka, r0 = sp.symbols("X Y Z x0 y0 c X0 Y0 Z0 omega phi kappa r0")
    dsyms = [sp.Symbol(f"d{j}") for j in range(len(dist))]
    co, so, cp, spn, ck, sk = sp.cos(om), sp.sin(om), sp.cos(ph), sp.sin(ph), sp.cos(ka), sp.sin(ka)
    r11, r12, r13 = cp * ck, -cp * sk, spn
    r21, r22, r23 = co * sk + so * spn * ck, co * ck - so * spn * sk, -so * cp
    r31, r32, r33 = so * sk - co * spn * ck, so * ck + co * spn * sk, co * cp
    dX, dY, dZ = X - X0, Y - Y0, Z - Z0
    kx = r11 * dX + r21 * dY + r31 * dZ
    ky = r12 * dX + r22 * dY + r32 * dZ
    N = r13 * dX + r23 * dY + r33 * dZ
    xs, ys = -c * kx / N, -c * ky / N
    u, v = sp.symbols("u v")
    dx, dy = 0, 0
    for (k, o), s in zip(dist, dsyms):
        if k == "AI":
            Ri = (u * u + v * v) ** o - (r0 * r0) ** o
            dx += u * s * Ri; dy += v * s * Ri
        elif k == "ZX":
            dx += s * zernike(o, u, v, r0)
        elif k == "ZY":
            dy += s * zernike(o, u, v, r0)
        elif k == "ZZ":
            Zf = zernike(o, u, v, r0)
            dx += s * sp.diff(Zf, u); dy += s * sp.diff(Zf, v)
    dx = dx.subs({u: xs, v: ys}, simultaneous=True) if dx != 0 else 0
    dy = dy.subs({u: xs, v: ys}, simultaneous=True) if dy != 0 else 0
    fx, fy = x0 + xs + dx, y0 + ys + dy
    params = [X, Y, Z, x0, y0, c, X0, Y0, Z0, om, ph, ka] + dsyms
    allsyms = params + [r0]
    exprs = [fx, fy] + [sp.diff(fx, p) for p in params] + [sp.diff(fy, p) for p in params]
    return sp.lambdify(allsyms, exprs, modules="mpmath"), len(params)


def main():
    rng = np.random.Generator(np.random.Philox(20260516))
    out = {"comment": "independent sympy/mpmath(50 digits) derivatives of the Zernike models (even radial orders); local order "
                      "X,Y,Z,x0,y0,c,X0,Y0,Z0,omega,phi,kappa,dist...", "sets": {}}
    for name, dist in SETS.items():
        f, npar = model(dist)
        cases = []
        while len(cases) < 16:
            X, Y, Z = rng.uniform(-1000, 1000), rng.uniform(-150, 150), rng.uniform(-1000, 1000)
            om, ph, ka = rng.uniform(-np.pi, np.pi), rng.uniform(-1.2, 1.2), rng.uniform(-np.pi, np.pi)
            co, so, cp, spn = np.cos(om), np.sin(om), np.cos(ph), np.sin(ph)
            r3 = np.array([spn, -so * cp, co * cp])
            st = np.array([X, Y, Z]) + r3 * rng.uniform(1500, 2500) + rng.normal(0, 250, 3)
            c, x0, y0, r0 = 28.78507 + rng.normal(0, 0.5), rng.normal(0, 0.05), rng.normal(0, 0.05), 13.488
            dvals = [rng.normal(0, 1) * (1e-4 if k == "AI" else 1e-3) for k, o in dist]
            args = [X, Y, Z, x0, y0, c, st[0], st[1], st[2], om, ph, ka] + dvals + [r0]
            vals = f(*[mp.mpf(float(a)) for a in args])
            fx, fy = float(vals[0]), float(vals[1])
            if abs(fx) > 18 or abs(fy) > 12 or math.hypot(fx, fy) < 1.0:
                continue
            xp, yp = fx + rng.normal(0, 5e-4), fy + rng.normal(0, 5e-4)
            w = [float(mp.mpf(xp) - vals[0]), float(mp.mpf(yp) - vals[1])]
            cases.append({"point": [X, Y, Z], "io": [x0, y0, c], "eo": [float(s) for s in st] + [om, ph, ka], "r0": r0,
                          "dist_values": dvals, "obs": [xp, yp], "w": w,
                          "Ax": [float(v) for v in vals[2:2 + npar]], "Ay": [float(v) for v in vals[2 + npar:2 + 2 * npar]]})
        out["sets"][name] = {"dist": [[KINDS[k], o] for k, o in dist], "cases": cases}
        print(name, len(cases))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "jacobian_rows_zernike.json")
    with open(path, "w") as fh:
        json.dump(out, fh)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
