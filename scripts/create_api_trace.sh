#!/bin/bash
# which HIP API calls engine creation spends its host time in:  bash scripts/create_api_trace.sh  (GPU box)
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"
cd /tmp; export TMPDIR=/tmp
rm -rf "$O/create_api"
rocprofv3 --hip-runtime-trace --stats --output-format csv -d "$O/create_api" -o t -- python3 "$R/scripts/create_time.py" cfg4 2 > "$O/create_api.log" 2>&1
python3 - "$O/create_api" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*hip_api_stats.csv") or glob.glob(sys.argv[1] + "/*/*hip_api_stats.csv")
print(f)
for r in sorted(csv.DictReader(open(f[0])), key=lambda r: -float(r["TotalDurationNs"]))[:18]:
    print(f'{r["Name"]:40s} calls {r["Calls"]:>7s} total {float(r["TotalDurationNs"]) / 1e6:9.1f} ms avg {float(r["AverageNs"]) / 1e3:9.1f} us')
PY
tail -3 "$O/create_api.log"
