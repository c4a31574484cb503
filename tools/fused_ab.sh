# scratch A/B script (edit per experiment)
R=$GRAFT_REPO_ROOT
for wl in cfg3 cfg2; do
  for v in "LIDARGS_MIX_FRONT=0 LIDARGS_TAIL_LIST=0" "LIDARGS_MIX_FRONT=1 LIDARGS_TAIL_LIST=0" "LIDARGS_MIX_FRONT=0 LIDARGS_TAIL_LIST=1" "LIDARGS_MIX_FRONT=1 LIDARGS_TAIL_LIST=1" "LIDARGS_MIX_FRONT=0 LIDARGS_TAIL_LIST=0" "LIDARGS_MIX_FRONT=1 LIDARGS_TAIL_LIST=1"; do
    echo "== $wl $v"; env $v python $R/tools/time_cfg.py $wl 2>&1 | tail -1 | cut -c1-330
  done
done
