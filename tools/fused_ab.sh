# scratch A/B script (edit per experiment)
R=$GRAFT_REPO_ROOT
for v in "LIDARGS_WORK_LISTS=0" "LIDARGS_WORK_LISTS=1" "LIDARGS_WORK_LISTS=0" "LIDARGS_WORK_LISTS=1"; do
  echo "== cfg5 $v"; env $v python $R/bench.py --workload cfg5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],4), d['stage_ms'])"
done
