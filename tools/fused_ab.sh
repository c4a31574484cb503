# scratch A/B script (edit per experiment)
R=$GRAFT_REPO_ROOT
for v in "LIDARGS_SMALL_SORT_MAX=16384" "LIDARGS_SMALL_SORT_MAX=6000" "LIDARGS_SMALL_SORT_MAX=2048" "LIDARGS_SMALL_SORT_MAX=0" "LIDARGS_SMALL_SORT_MAX=16384" "LIDARGS_SMALL_SORT_MAX=2048"; do
  echo "== $v"
  env $v python $R/bench.py --workload cfg1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('eager', round(d['ms_per_step'],4), d['stage_ms'])"
  env $v python $R/bench.py --workload cfg1 --graph --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('graph', round(d['ms_per_step'],4))"
done
