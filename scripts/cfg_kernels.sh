#!/bin/bash
# per-kernel time of the LM passes of one config:  bash scripts/cfg_kernels.sh cfg3 [steps]   (GPU box)
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"; CFG=${1:-cfg3}; STEPS=${2:-40}
cd /tmp; export TMPDIR=/tmp
rm -rf "$O/ck_$CFG"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/ck_$CFG" -o t -- python3 "$R/bench.py" --config $CFG --steps $STEPS --warmup 2 --iterations-only --no-cpu-baseline --no-dense-mode > "$O/ck_$CFG.log" 2>&1
python3 - "$O/ck_$CFG/t_kernel_stats.csv" $STEPS <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) + 2
tot = 0.0
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    per = float(r["TotalDurationNs"]) / steps / 1000.0
    if per > 2.0:
        print(f'{r["Name"][:70]:70s} calls/pass {int(r["Calls"]) / steps:6.2f}  avg {float(r["AverageNs"]) / 1000:8.1f} us  per pass {per:8.1f} us')
    tot += per
print("sum per pass", round(tot, 1), "us")
PY
