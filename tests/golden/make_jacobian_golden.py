"""Generates tests/golden/jacobian_rows.json: independent golden vectors for residual + Jacobian rows.

The vectors are NOT produced by the build's own C/HIP code: the model function
    x = x0 + xs + sum(dx(xs, ys, N)),   y = y0 + ys + sum(dy(xs, ys, N))
(xs, ys, N from PartialDerivativeFactory.java:137-152; dx,dy from RadiallySymmetricDistortionModelFactory.java:58-63,
TangentialDistortionModelFactory.java:55-56,76-82, AffinityShearDistortionModelFactory.java:45-46,
RadialDistanceDistortionModelFactory.java:58-63) is differentiated SYMBOLICALLY with sympy and evaluated with mpmath
at 50 digits.  Run once:  python tests/golden/make_jacobian_golden.py
"""
import json
import os

import mpmath as mp
import numpy as np
import sympy as sp

mp.mp.dps = 50

KINDS = {"CX": 0, "CY": 1, "BX": 2, "BY": 3, "BI": 4, "AI": 5, "DI": 6}
SETS = {
    "pinhole": [],
    "radial": [("AI", 1), ("AI", 2), ("AI", 3)],
    "full": [("CX", 0), ("CY", 0), ("BX", 0), ("BY", 0), ("BI", 1), ("AI", 1), ("AI", 2), ("AI", 3),
             ("DI", 1), ("DI", 2), ("DI", 3)],
    "tangential2": [("BX", 0), ("BY", 0), ("BI", 1), ("BI", 2), ("AI", 2)],
}
SCALE = {"CX": 1e-4, "CY": 1e-4, "BX": 1e-5, "BY": 1e-5, "BI": 1e-4, "AI": None, "DI": None}


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
    r2 = xs * xs + ys * ys
    dx, dy = 0, 0
    val = {}
    for (k, o), s in zip(dist, dsyms):
        val[(k, o)] = s
    if ("CX", 0) in val:
        dx += val[("CX", 0)] * xs + val[("CY", 0)] * ys
    if ("BX", 0) in val:
        bx, by = val[("BX", 0)], val[("BY", 0)]
        S = 1
        for (k, o), s in zip(dist, dsyms):
            if k == "BI":
                S = S + s * r2 ** o
        dx += (bx * (r2 + 2 * xs * xs) + by * 2 * xs * ys) * S
        dy += (by * (r2 + 2 * ys * ys) + bx * 2 * xs * ys) * S
    for (k, o), s in zip(dist, dsyms):
        if k == "AI":
            Ri = r2 ** o - (r0 * r0) ** o
            dx += xs * s * Ri; dy += ys * s * Ri
        elif k == "DI":
            Ri = r2 ** o - (r0 * r0) ** o
            dx += xs * s * Ri / N; dy += ys * s * Ri / N
    fx, fy = x0 + xs + dx, y0 + ys + dy
    params = [X, Y, Z, x0, y0, c, X0, Y0, Z0, om, ph, ka] + dsyms
    allsyms = params + [r0]
    exprs = [fx, fy] + [sp.diff(fx, p) for p in params] + [sp.diff(fy, p) for p in params]
    f = sp.lambdify(allsyms, exprs, modules="mpmath")
    return f, len(params)


def main():
    rng = np.random.Generator(np.random.Philox(20260515))
    out = {"comment": "independent sympy/mpmath(50 digits) derivatives; local order X,Y,Z,x0,y0,c,X0,Y0,Z0,omega,phi,kappa,dist...",
           "sets": {}}
    for name, dist in SETS.items():
        f, npar = model(dist)
        cases = []
        while len(cases) < 32:
            # a station 1.5-2.5 m from the point, looking roughly at it
            X, Y, Z = rng.uniform(-1000, 1000), rng.uniform(-150, 150), rng.uniform(-1000, 1000)
            om, ph, ka = rng.uniform(-np.pi, np.pi), rng.uniform(-1.2, 1.2), rng.uniform(-np.pi, np.pi)
            co, so, cp, spn = np.cos(om), np.sin(om), np.cos(ph), np.sin(ph)
            r3 = np.array([spn, -so * cp, co * cp])
            dist_cam = rng.uniform(1500, 2500)
            off = rng.normal(0, 250, 3)
            st = np.array([X, Y, Z]) + r3 * dist_cam + off
            c, x0, y0, r0 = 28.78507 + rng.normal(0, 0.5), rng.normal(0, 0.05), rng.normal(0, 0.05), 13.488
            dvals = []
            for k, o in dist:
                if k == "AI":
                    dvals.append(rng.normal(0, 1) * [1e-4, 1e-7, 1e-10][o - 1])
                elif k == "DI":
                    dvals.append(rng.normal(0, 1) * [1e-3, 1e-6, 1e-9][o - 1])
                else:
                    dvals.append(rng.normal(0, 1) * SCALE[k])
            args = [X, Y, Z, x0, y0, c, st[0], st[1], st[2], om, ph, ka] + dvals + [r0]
            vals = f(*[mp.mpf(float(a)) for a in args])
            fx, fy = float(vals[0]), float(vals[1])
            if abs(fx) > 18 or abs(fy) > 12:
                continue
            xp, yp = fx + rng.normal(0, 5e-4), fy + rng.normal(0, 5e-4)
            w = [float(mp.mpf(xp) - vals[0]), float(mp.mpf(yp) - vals[1])]
            Ax = [float(v) for v in vals[2:2 + npar]]
            Ay = [float(v) for v in vals[2 + npar:2 + 2 * npar]]
            cases.append({"point": [X, Y, Z], "io": [x0, y0, c], "eo": [float(s) for s in st] + [om, ph, ka],
                          "r0": r0, "dist_values": dvals, "obs": [xp, yp], "w": w, "Ax": Ax, "Ay": Ay})
        out["sets"][name] = {"dist": [[KINDS[k], o] for k, o in dist], "cases": cases}
        print(name, len(cases))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "jacobian_rows.json")
    with open(path, "w") as fh:
        json.dump(out, fh)
    print("wrote", path, os.path.getsize(path))


if __name__ == "__main__":
    main()
