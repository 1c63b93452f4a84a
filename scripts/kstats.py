"""Print the top of a rocprofv3 kernel_stats CSV."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 24]:
    print(r['Name'][:60].ljust(60), r['Calls'].rjust(6), f"{float(r['TotalDurationNs'])/1e6:9.2f}ms", f"{float(r['AverageNs'])/1e3:9.1f}us")
