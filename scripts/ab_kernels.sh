cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ab_prof -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 1 --iterations-only > $GRAFT_REPO_ROOT/gpurun_out/ab_prof.log 2>&1
python3 - <<'PY'
import csv,glob,os
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/ab_prof/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if any(k in n for k in ('blk_T','blk_pp','blk_pc','blk_cc','blk_elim','blk_tfix','chol_tile')): print(n[:60], r['Calls'], r['AverageNs'])
PY
