#!/bin/bash
# HIP API calls longer than 5 ms of a python script:  bash scripts/api_long_calls.sh scripts/two_engines.py   (GPU box)
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"
cd /tmp; export TMPDIR=/tmp
rm -rf "$O/api_long"
rocprofv3 --hip-runtime-trace --output-format csv -d "$O/api_long" -o t -- python3 "$R/$1" > "$O/api_long.log" 2>&1
python3 - "$O/api_long/t_hip_api_trace.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if d > 5.0:
        print(f'{(int(r["Start_Timestamp"]) - t0) / 1e6:10.1f} ms  {r["Function"]:36s} {d:8.1f} ms')
PY
grep -E "^e|close" "$O/api_long.log" | cut -c1-200
