"""Collects the per-dispatch counters of scripts/gather_counters.sh into one table: mean per launch of the LM-pass kernels named below."""
import csv, glob, json, os, sys

KERNELS = ["blk_pp_gather_kernel", "blk_T_mfma_kernel", "blk_pc_gather_kernel", "blk_cc_kernel"]


def main():
    out_dir, tag, nsets = sys.argv[1], sys.argv[2], int(sys.argv[3])
    table = {k: {} for k in KERNELS}
    for i in range(1, nsets + 1):
        fs = glob.glob(os.path.join(out_dir, f"gc_{tag}_{i}", "*", "*counter_collection.csv"))
        if not fs:
            continue
        acc = {}
        for r in csv.DictReader(open(fs[0])):
            for k in KERNELS:
                if k in r["Kernel_Name"]:
                    acc.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
        for (k, c), v in acc.items():
            table[k][c] = sum(v) / len(v)
            table[k]["dispatches"] = len(v)
    print(json.dumps(table, indent=1))


if __name__ == "__main__":
    main()
