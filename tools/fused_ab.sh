# scratch A/B script (edit per experiment): env variants x workloads through tools/time_cfg.py (stage times with per-frame events)
#   gpurun -- 'cd $GRAFT_REPO_ROOT && bash tools/fused_ab.sh'
R=$GRAFT_REPO_ROOT
for wl in cfg3 cfg2 cfg4; do
  for v in "LIDARGS_WORK_LISTS=0" "LIDARGS_WORK_LISTS=1"; do
    echo "== $wl $v"; env $v python $R/tools/time_cfg.py $wl 2>&1 | tail -1 | cut -c1-330
  done
done
