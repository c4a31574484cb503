# A/B of environment settings over workloads, per-stage times (tools/time_cfg.py; cfg5 through bench.py's stage events).
#   gpurun -- 'bash tools/ab_env.sh "LIDARGS_EXACT_FLAGS=0 LIDARGS_EXACT_FLAGS=1" "cfg3 cfg2 cfg4 cfg5"'
R=${GRAFT_REPO_ROOT:-/root/repo}
VARIANTS=${1:-"X=0"}
WLS=${2:-"cfg3"}
REPS=${3:-2}
for rep in $(seq $REPS); do
for wl in $WLS; do
  for v in $VARIANTS; do
    echo "== $wl $v"
    if [ "$wl" = cfg5 ]; then env $v bash $R/tools/stage_ms.sh cfg5 100 2>&1 | tail -1 | cut -c1-600
    else env $v python $R/tools/time_cfg.py $wl 2>&1 | tail -1 | cut -c1-600; fi
  done
done
done
