#!/bin/bash
# bench line + rocprofv3 kernel stats + the two PMC passes for profiles/ (run on the GPU box through gpurun)
set -e
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"; TAG=${1:-x}
cd /tmp; export TMPDIR=/tmp
python "$R/bench.py" --steps 10 --warmup 2 > "$O/bench_$TAG.json" 2> "$O/bench_$TAG.err"
cat "$O/bench_$TAG.json"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_$TAG" -o t -- python "$R/bench.py" --steps 3 --warmup 1 --iterations-only > "$O/prof_$TAG.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmc_fetch_$TAG" -- python "$R/bench.py" --steps 2 --warmup 0 --iterations-only > "$O/pmc_fetch_$TAG.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmc_write_$TAG" -- python "$R/bench.py" --steps 2 --warmup 0 --iterations-only > "$O/pmc_write_$TAG.log" 2>&1
ls "$O/pmc_fetch_$TAG" "$O/pmc_write_$TAG"
