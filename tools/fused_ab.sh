# scratch A/B script (edit per experiment)
R=$GRAFT_REPO_ROOT
for v in "LIDARGS_NG_ROWS=2" "LIDARGS_NG_ROWS=1" "LIDARGS_NG_ROWS=2" "LIDARGS_NG_ROWS=1"; do
  echo "== $v"
  env $v python $R/bench.py --workload decode --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('decode', round(d['ms_per_step'],4), d.get('parts') or d.get('stage_ms') or '')"
  env $v python $R/bench.py --workload train_step --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('train_step', round(d['ms_per_step'],4), d.get('parts') or d.get('stage_ms') or '')"
done
