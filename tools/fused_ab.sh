R=$GRAFT_REPO_ROOT
for wl in cfg3 cfg2 cfg4; do
  for v in "LIDARGS_TILE_KEY32=1" "LIDARGS_TILE_KEY32=0"; do
    echo "== $wl $v"; env $v python $R/tools/time_cfg.py $wl 2>&1 | tail -1 | cut -c1-230
  done
done
