R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/tr_small -o t -- python $R/bench.py --workload cfg1 --no-cpu-baseline --steps 200 --warmup 20 > /dev/null 2>&1
DB=$(ls /tmp/tr_small/*/*.db /tmp/tr_small/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_stats.py $DB | head -8 | python -c "import sys,csv
for r in csv.reader(sys.stdin): print(r[0][:50], r[1], r[3])"
