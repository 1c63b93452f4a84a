"""(b)/(a) durations of one factorisation from a rocprofv3 kernel trace (exploration helper)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'rows_kernel' in r['Kernel_Name']]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 2
it = rows[idx[which]:idx[which + 1]]
dur = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
f = [i for i, r in enumerate(it) if 'potrf_diag' in r['Kernel_Name']]
seg = it[f[0]:f[-1] + 1]
span = (int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e3
big = [r for r in seg if '128, 128' in r['Kernel_Name'] and int(r['Grid_Size_X']) // 256 > 300]
print('span', round(span), 'sum(b)', round(sum(map(dur, big))), 'n', len(big))
for r in big[:12]:
    wg = int(r['Grid_Size_X']) // 256
    print(f"  wg={wg:5d} dur={dur(r):8.1f} q{r['Queue_Id']}")
diag = [r for r in seg if 'potrf_diag' in r['Kernel_Name']]
print('diag n', len(diag), 'avg', sum(map(dur, diag)) / len(diag))
