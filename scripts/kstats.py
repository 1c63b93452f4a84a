"""Per-kernel table of the LM passes of a rocprofv3 --kernel-trace run of bench.py: dispatches from the first rows_kernel on
(engine creation factors 500 small dispersion blocks with the same kernels and would swamp the averages).
python scripts/kstats.py <kernel_trace.csv> [out.csv]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = next(i for i, r in enumerate(rows) if "rows_kernel" in r["Kernel_Name"])
n_pass = sum("rows_kernel" in r["Kernel_Name"] for r in rows)
agg = collections.OrderedDict()
for r in rows[first:]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(r["Kernel_Name"], [0, 0.0, 1e30, 0.0])
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
out = [("kernel", "calls", "calls_per_pass", "total_us", "avg_us", "min_us", "max_us", "us_per_pass")]
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    out.append((k, a[0], round(a[0] / n_pass, 2), round(a[1], 1), round(a[1] / a[0], 1), round(a[2], 1), round(a[3], 1), round(a[1] / n_pass, 1)))
w = csv.writer(open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout)
w.writerow(("# LM passes in the trace", n_pass))
w.writerows(out)
