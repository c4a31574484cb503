R=$GRAFT_REPO_ROOT
for wl in cfg3 cfg2; do
  for v in "X=0" ; do
    echo "== $wl $v"; env $v python $R/tools/time_cfg.py $wl 2>&1 | tail -1 | cut -c1-330
  done
done
