R=$GRAFT_REPO_ROOT
for wl in cfg3 cfg2; do
  for v in "LIDARGS_FUSED=0" "LIDARGS_FUSED=1" "LIDARGS_FUSED=1 LIDARGS_FUSED_MAXROUNDS=1" "LIDARGS_FUSED=1 LIDARGS_FUSED_MAXROUNDS=2" "LIDARGS_FUSED=1 LIDARGS_FUSED_MAXROUNDS=1 LIDARGS_FUSED_WAVES=4" "LIDARGS_FUSED=1 LIDARGS_FUSED_MAXROUNDS=1 LIDARGS_FUSED_WAVES=16"; do
    echo "== $wl $v"; env $v python $R/tools/time_cfg.py $wl 2>&1 | tail -1 | sed 's/.*render_fused/render_fused/; s/.*render_pass1/render_pass1/'
  done
done
