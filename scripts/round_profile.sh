#!/bin/bash
# bench line + rocprofv3 kernel stats + the two PMC passes for profiles/ (run on the GPU box through gpurun):
#   bash scripts/round_profile.sh <tag>        -> gpurun_out/{bench,prof,pmc_fetch,pmc_write}_<tag>*; profiles/pmc_traffic.json rewritten
set -e
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"; TAG=${1:-x}
cd /tmp; export TMPDIR=/tmp
python3 "$R/bench.py" --steps 10 --warmup 2 > "$O/bench_$TAG.json" 2> "$O/bench_$TAG.err"
cat "$O/bench_$TAG.json"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_$TAG" -o t -- python3 "$R/bench.py" --steps 3 --warmup 1 --iterations-only > "$O/prof_$TAG.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_dm_$TAG" -o t -- python3 "$R/scripts/dense_mode_pass.py" > "$O/prof_dm_$TAG.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmc_fetch_$TAG" -- python3 "$R/bench.py" --steps 2 --warmup 0 --iterations-only > "$O/pmc_fetch_$TAG.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmc_write_$TAG" -- python3 "$R/bench.py" --steps 2 --warmup 0 --iterations-only > "$O/pmc_write_$TAG.log" 2>&1
python3 "$R/scripts/pmc_traffic.py" "$O/pmc_fetch_$TAG" "$O/pmc_write_$TAG" "$TAG" > "$O/pmc_traffic_$TAG.log" 2>&1
cp "$R/profiles/pmc_traffic.json" "$O/pmc_traffic_$TAG.json"
# the smaller BASELINE configs and the strips scene on the same build (bench lines only)
for c in cfg2 cfg3 cfg4_local; do
  python3 "$R/bench.py" --config $c --steps 50 --warmup 5 --no-cpu-baseline --no-dense-mode > "$O/bench_${c}_$TAG.json" 2> "$O/bench_${c}_$TAG.err"
done
# engine creation (2nd / 3rd engine of a process) and the accuracy against the extended-precision truth
python3 "$R/scripts/create_time.py" cfg4 > "$O/create_time_$TAG.log" 2>&1
python3 "$R/scripts/exactN_compare.py" cfg3b "$O/exactN_cfg3b_$TAG.json" > "$O/exactN_cfg3b_$TAG.log" 2>&1
python3 "$R/scripts/exactN_compare.py" cfg4 "$O/exactN_cfg4_$TAG.json" > "$O/exactN_cfg4_$TAG.log" 2>&1
find "$O/prof_$TAG" "$O/prof_dm_$TAG" -name "*kernel_stats.csv" | head
