#!/usr/bin/env python
"""Device against the GROUND TRUTH (tests/golden/<cfg>/<cfg>_exactN*.npz: normal equations assembled in extended precision and solved
exactly, tests/golden/make_exactN.py), beside the oracle's (= the reference algorithm's) own error against the same truth.

    python scripts/exactN_compare.py cfg3b|cfg4 [out.json]        (GPU box)

Per inversion mode (FULL = expanded from the reduced inverse, REDUCED): Qxx over the fixture's 400 sample columns, correlation-scaled
(|dQ_ij| / sqrt(Q_ii Q_jj)), and the sampled variances; N.v of the device's assembled system; the first rows of the device's weights
sigma0^2 inv(D) of image blocks 0 and 1."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bundle_adjustment_amd import engine, scene  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")


def packed_matvec(ap, v):
    y = np.zeros(v.size); off = 0
    for r in range(v.size):
        row = ap[off:off + r + 1]
        y[r] += row @ v[:r + 1]; y[:r] += row[:r] * v[r]
        off += r + 1
    return y


def scaled(Q, ref):
    sd = np.sqrt(np.abs(np.diag(ref)))
    return float((np.abs(Q - ref) / np.outer(sd, sd)).max())


def main():
    cfg = sys.argv[1]
    fp = scene.config({"cfg3b": "cfg3_block", "cfg4": "cfg4"}[cfg])
    t = np.load(os.path.join(G, cfg, f"{cfg}_exactN.npz"))
    tm = json.load(open(os.path.join(G, cfg, f"{cfg}_exactN.json")))
    z = np.load(os.path.join(G, cfg, f"{cfg}_converged.npz"))
    meta = json.load(open(os.path.join(G, cfg, f"{cfg}_converged.json")))
    cols = t["sample_cols"].astype(np.int64)
    out = {"config": cfg, "oracle_vs_truth": {"Qsample": tm["oracle_Qsample_err"], "variances": tm["oracle_diag_err"],
                                              "Nv": tm["oracle_Nv_err"], "inv_D_blocks": tm["oracle_P_err_blocks"]}}
    s2 = fp.sigma2apriori
    probe = np.random.Generator(np.random.Philox(meta["probe_seed"])).standard_normal(fp.n_unknowns)
    for det in (0, 1):
        eng = engine.Engine(fp, deterministic=det)
        eng.set_parameters(z["values"])
        for name, inv in (("FULL", engine.INVERT_FULL_EXPANDED), ("REDUCED", engine.INVERT_REDUCED)):
            eng.prepare_inverse(inv)
            eng.build(s2, 0.0)
            eng.solve(inv)
            k = eng.cofactor_order()
            keep = cols < k
            Qs = eng.get_cofactor_sub(cols[keep].astype(np.int32))
            truth = t["Qsample_true"][np.ix_(keep, keep)]
            orac = z["Qsample"][np.ix_(keep, keep)]
            r = {"device_vs_truth": scaled(Qs, truth), "device_vs_oracle": scaled(Qs, orac), "oracle_vs_truth": scaled(orac, truth),
                 "device_variances_vs_truth": float(np.abs(np.diag(Qs) / np.diag(truth) - 1).max()),
                 "oracle_variances_vs_truth": float(np.abs(np.diag(orac) / np.diag(truth) - 1).max())}
            out[f"{name}_det{det}"] = r
            print(name, "deterministic (default)" if det else "arrival-order sums", json.dumps(r), flush=True)
        if det == 0:
            eng.prepare_inverse(engine.INVERT_FULL)
            eng.build(s2, 0.0)
            N, n = eng.get_normal()
            Nv = packed_matvec(N, probe)
            del N
            out["device_Nv_vs_truth"] = float(np.abs(Nv - t["Nv_exact"]).max() / np.abs(t["Nv_exact"]).max())
            print("N.v device vs truth", out["device_Nv_vs_truth"], " oracle vs truth", tm["oracle_Nv_err"], flush=True)
            if hasattr(eng, "get_block_weight"):
                pe = []
                for b in range(t["P_rows_exact"].shape[0]):
                    Pd = eng.get_block_weight(b) * s2
                    ref = t["P_rows_exact"][b]
                    pe.append(float(np.abs(Pd[:ref.shape[0]] - ref).max() / np.abs(ref).max()))
                out["device_inv_D_blocks_vs_truth"] = pe
                print("sigma0^2 inv(D), first rows of blocks 0, 1: device vs truth", pe, " oracle vs truth", tm["oracle_P_err_blocks"][:2], flush=True)
        eng.close()
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
