#!/bin/bash
# per-kernel durations of the substitution chains in the LM pass (run on the GPU box):  bash scripts/prof_chain.sh <tag>
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"; TAG=${1:-x}
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_$TAG" -o t -- python3 "$R/bench.py" --steps 3 --warmup 1 --iterations-only > "$O/prof_$TAG.log" 2>&1
python3 - "$O/prof_$TAG/t_kernel_stats.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if 'chain' in n or 'symv' in n or '<0, 1, 128' in n:
        print(n[:60], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
