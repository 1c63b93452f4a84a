"""Generates tests/golden/<cfg>/<cfg>_weight_certificate.json: a binary128 certificate for EVERY weight matrix of a configuration.

For every jointly dispersed image group: P = sigma0^2 inv(D) in extended precision + one compensated Newton step
(oracle/ba_exact.c, oracle_exact_block_weight -- what the truth fixtures *_exactN.npz are assembled from) and the fp64 weight of the
reference's dpptrf + dpptri (DOPG:82-86, oracle_block_weight), each probed with 4 random sign vectors in binary128
(oracle_inverse_residual_probe_q: max_i |(v - D P v / sigma0^2)_i|, covers every row of the matrix).  The truth fixtures' own certificate
(make_exactN.py) is exact but looks at 32 rows of block 0 only; this one looks at all blocks.
  python tests/golden/make_weight_certificate.py cfg4        (8 processes: ~6 minutes)
"""
import json
import multiprocessing as mp
import os
import sys
import time

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
import numpy as np  # noqa: E402

from bundle_adjustment_amd import scene  # noqa: E402
from oracle import oracle as om  # noqa: E402

NVEC = 4
_fp = None
_o = None


def one(b):
    fp, o = _fp, _o
    s2 = fp.sigma2apriori
    L = om.lib()
    Ph, Pl = o.exact_block_weight(s2, b)
    P64 = o.block_weight(s2, b)
    m = Ph.shape[0]
    D = np.ascontiguousarray(fp.blk_disp[fp.blk_disp_offset[b]:fp.blk_disp_offset[b] + m * m])
    re = L.oracle_inverse_residual_probe_q(m, om._p(D), om._p(Ph), om._p(Pl), s2, NVEC, b + 1)
    rf = L.oracle_inverse_residual_probe_q(m, om._p(D), om._p(P64), None, s2, NVEC, b + 1)
    return b, m, float(re), float(rf), float(np.abs(P64 - Ph).max() / np.abs(Ph).max())


def main():
    global _fp, _o
    cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    os.environ["OMP_NUM_THREADS"] = "1"
    t0 = time.time()
    _fp = scene.config(cfg)
    _o = om.Oracle(_fp)
    nb = _fp.n_image_blocks
    print(f"{cfg}: {nb} blocks, scene in {time.time() - t0:.0f} s", flush=True)
    with mp.get_context("fork").Pool(procs) as pool:
        res = []
        for r in pool.imap_unordered(one, range(nb), chunksize=1):
            res.append(r)
            if len(res) % 50 == 0:
                print(f"  {len(res)} / {nb}  ({time.time() - t0:.0f} s)", flush=True)
    res.sort()
    re = np.array([r[2] for r in res]); rf = np.array([r[3] for r in res]); er = np.array([r[4] for r in res])
    out = {
        "config": cfg, "blocks": nb, "probe_vectors": NVEC,
        "what": "max_i |(v - D P v / sigma0^2)_i| over random sign vectors v, binary128; P = extended-precision weight (hi + lo) / fp64 dpptrf + dpptri weight",
        "order_min": int(min(r[1] for r in res)), "order_max": int(max(r[1] for r in res)),
        "residual_exact_max": float(re.max()), "residual_exact_median": float(np.median(re)),
        "residual_fp64_max": float(rf.max()), "residual_fp64_median": float(np.median(rf)), "residual_fp64_min": float(rf.min()),
        "fp64_weight_error_max": float(er.max()), "fp64_weight_error_median": float(np.median(er)),
        "per_block_residual_exact": [float(f"{x:.3e}") for x in re],
        "per_block_residual_fp64": [float(f"{x:.3e}") for x in rf],
    }
    path = os.path.join(here, cfg, f"{cfg}_weight_certificate.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if not k.startswith("per_block")}, indent=1))
    print(f"wrote {path} in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
