#!/bin/bash
# Hardware counters of the point x point gather (blk_pp_gather_kernel) and blk_T at BASELINE config 4: which pipe of a CU the kernel
# occupies (vector memory / LDS / VALU), LDS bank conflicts of the strip's atomic adds, L2 hit rate and what goes behind L2.
#   bash scripts/gather_counters.sh <tag>  (GPU box)  -> gpurun_out/gc_<tag>_<set>/..., summary gpurun_out/gather_counters_<tag>.json
# One rocprofv3 pass per counter set (slot limits per block); counters only, no tracing besides --kernel-trace.
set -e
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"; TAG=${1:-x}; CFG=${2:-cfg4}
cd /tmp; export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL" \
           "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$O/gc_${TAG}_$i" -- python3 "$R/bench.py" --config $CFG --steps 2 --warmup 0 --iterations-only > "$O/gc_${TAG}_$i.log" 2>&1
  echo "set $i done"
done
python3 "$R/scripts/gather_counters.py" "$O" "$TAG" 6 > "$O/gather_counters_$TAG.json"
cat "$O/gather_counters_$TAG.json"
