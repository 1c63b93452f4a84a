"""Parses rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of bench.py and writes profiles/pmc_traffic.json.

HBM bytes per launch of the dominant kernel (the lower-triangular Cholesky trailing update, gemm_f64_kernel<0,0,128,128,1>) = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE/WRITE_SIZE are in KiB and FETCH_SIZE reports half of
the bytes of wide coalesced reads on gfx950 (MI355X_MICROARCH.md, HBM section); the two counters come from separate
passes (TCC slot limit).  Usage: python scripts/pmc_traffic.py <fetch_dir> <write_dir>"""
import csv, glob, json, os, sys

def dispatches(d, counter):
    f = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))[0]
    out = []
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and "gemm_f64_kernel<0, 0, 128, 128, 1>" in r["Kernel_Name"]:
            out.append((int(r["Grid_Size"]) // 256, float(r["Counter_Value"])))
    return out

def main():
    fetch_dir, write_dir = sys.argv[1], sys.argv[2]
    # the trailing update has its own kernel symbol (gemm_f64.h, TAG 1): every dispatch of it counts
    fe = [v for g, v in dispatches(fetch_dir, "FETCH_SIZE")]
    wr = [v for g, v in dispatches(write_dir, "WRITE_SIZE")]
    n = min(len(fe), len(wr))
    fetch_kb, write_kb = sum(fe) / len(fe), sum(wr) / len(wr)
    out = {"kernel": "gemm_f64_kernel<0, 0, 128, 128, 1> (Cholesky trailing update), all dispatches of the run",
           "dispatches_matched": n, "fetch_size_kib_per_launch_raw": fetch_kb, "write_size_kib_per_launch": write_kb,
           "hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
           "hbm_bytes_per_launch_uncorrected": (fetch_kb + write_kb) * 1024.0,
           "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B for wide coalesced reads); "
                   "the C-tile loads are 8 B/lane, for which the guide gives no calibration, so the corrected figure is an upper bound"}
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))

if __name__ == "__main__":
    main()
