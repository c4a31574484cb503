# kernel times and SQ counters of the 10 k-Gaussian frame (bench.py --workload cfg1):  bash tools/trace_small.sh
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/tr_small -o t -- python $R/bench.py --workload cfg1 --no-cpu-baseline --steps 200 --warmup 20 > /dev/null 2>&1
DB=$(ls /tmp/tr_small/*/*.db /tmp/tr_small/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_stats.py $DB | head -14 | python -c "import sys,csv
for r in csv.reader(sys.stdin): print(r[0][:50], r[1], r[3])"
SQ="SQ_WAVES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"
rocprofv3 --kernel-trace --pmc $SQ -d /tmp/sq_small -o t -- python $R/bench.py --workload cfg1 --no-cpu-baseline --steps 20 --warmup 5 > /dev/null 2>&1
DB=$(ls /tmp/sq_small/*/*.db /tmp/sq_small/*.db 2>/dev/null | head -1)
python $R/tools/pmc_sq.py cfg1 "x" $DB | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d['kernels'].items():
    if 'small' in k or 'fused' in k or 'backward' in k: print(k[:60], v)"
