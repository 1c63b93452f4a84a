"""Parses rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of bench.py and writes profiles/pmc_traffic.json.

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE / WRITE_SIZE are in KiB, FETCH_SIZE reports half of
the bytes of wide coalesced reads on gfx950 (MI355X_MICROARCH.md, HBM section), the two counters come from separate passes
(TCC slot limit).  The dominant kernel (bench.py's `roofline`) first, then every kernel of the assembly with the
algorithmic bytes of DESIGN.md section 4 beside the counters.
Usage: python scripts/pmc_traffic.py <fetch_dir> <write_dir> [tag]"""
import csv, glob, json, os, sys, time

DOMINANT = "chol_tile_kernel"
ASSEMBLY = ["rows_kernel", "blk_T_mfma_kernel", "blk_T_kernel", "blk_elim_kernel", "blk_tfix_kernel", "blk_cc_kernel", "blk_pc_gather_kernel",
            "blk_pp_gather_kernel", "zero_lower_kernel", "direct_kernel"]
# algorithmic bytes per launch at config 4 (500 images, m = 1000 rows per image, reduced order 15014): DESIGN.md section 4
M2 = 500 * 1000.0 ** 2
ALGO = {"blk_T_kernel": 8 * M2, "blk_T_mfma_kernel": 8 * M2, "blk_pp_gather_kernel": 4 * M2 + 4 * 15014.0 ** 2, "rows_kernel": 250000 * 460.0,
        "zero_lower_kernel": 0.0}


def dispatches(d, counter):
    f = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))[0]
    out = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        for k in [DOMINANT] + ASSEMBLY:
            if k in name and not (k == "rows_kernel" and "slice_rows_kernel" in name):      # batchinv.hip's slicing kernel is not the Jacobian-rows kernel
                out.setdefault(k, []).append(float(r["Counter_Value"]))
    return out


def main():
    fetch_dir, write_dir = sys.argv[1], sys.argv[2]
    fe, wr = dispatches(fetch_dir, "FETCH_SIZE"), dispatches(write_dir, "WRITE_SIZE")
    def big(v):      # the dominant kernel also factors the 500 small dispersion blocks at create(): only the large dispatches count
        return [x for x in v if x >= 0.5 * max(v)] if v else v
    def rec(k):
        if k == DOMINANT:
            fe[k], wr[k] = big(fe.get(k, [])), big(wr.get(k, []))
        f = sum(fe[k]) / len(fe[k]) if fe.get(k) else None
        w = sum(wr[k]) / len(wr[k]) if wr.get(k) else None
        if f is None or w is None:
            return None
        r = {"dispatches": min(len(fe[k]), len(wr[k])), "fetch_size_kib_per_launch_raw": f, "write_size_kib_per_launch": w,
             "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0, "hbm_bytes_per_launch_uncorrected": (f + w) * 1024.0}
        if k in ALGO and ALGO[k] > 0:
            r["algorithmic_bytes_per_launch"] = ALGO[k]
            r["counter_to_algorithmic"] = r["hbm_bytes_per_launch"] / ALGO[k]
        return r
    dom = rec(DOMINANT)
    out = {"kernel": DOMINANT + "<1, true> (dataflow Cholesky tile kernel, the whole factorisation in one launch; under counter collection "
                     "every dispatch is serialised, so the one-kernel form with inline diagonal blocks runs: same tile traffic as the product "
                     "form <1, false>, plus ~0.1 MB per diagonal block of work arrays), the factorisations of the timed passes",
           "collected": time.strftime("%Y-%m-%d %H:%M:%S"), "tag": sys.argv[3] if len(sys.argv) > 3 else "",
           "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B for wide coalesced reads); "
                   "8-B-per-lane loads (C tiles, gathers) are uncalibrated there, so the corrected figure is an upper bound"}
    if dom:
        out.update({k: dom[k] for k in dom})
    out["assembly_kernels"] = {k: rec(k) for k in ASSEMBLY if rec(k)}
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
