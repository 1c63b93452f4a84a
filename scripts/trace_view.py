"""Print the kernel timeline of one LM pass from a rocprofv3 kernel trace CSV (exploration helper)."""
import csv, sys
path = sys.argv[1]; which = int(sys.argv[2]) if len(sys.argv) > 2 else 3
lo = float(sys.argv[3]) if len(sys.argv) > 3 else 0; hi = float(sys.argv[4]) if len(sys.argv) > 4 else 1e18
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'rows_kernel' in r['Kernel_Name']]
it = rows[idx[which]:idx[which + 1]] if which + 1 < len(idx) else rows[idx[which]:]
t0 = int(it[0]['Start_Timestamp'])
for r in it:
    s = (int(r['Start_Timestamp']) - t0) / 1e3; e = (int(r['End_Timestamp']) - t0) / 1e3
    if s < lo or s > hi: continue
    n = r['Kernel_Name'].replace('jaicov::', '').replace('void ', '')[:40]
    print(f"{s:10.1f} {e:10.1f} {e-s:8.1f} q{r['Queue_Id']} {n:42s} wg={int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])}x{r['Grid_Size_Y']}")
