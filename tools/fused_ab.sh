# scratch A/B script (edit per experiment)
R=$GRAFT_REPO_ROOT
for wl in cfg3 cfg2 cfg4; do
  for v in "LIDARGS_XCD_STRIP=0" "LIDARGS_XCD_STRIP=1" "LIDARGS_XCD_STRIP=2" "LIDARGS_XCD_STRIP=4" "LIDARGS_XCD_STRIP=8" "LIDARGS_XCD_STRIP=0"; do
    echo "== $wl $v (slot-grid backward)"; env $v LIDARGS_WORK_LISTS=0 python $R/tools/time_cfg.py $wl 2>&1 | tail -1 | grep -o "ms/frame [0-9.]*\|'render_pass1': [0-9.]*\|'render_pass2': [0-9.]*\|'render_bwd': [0-9.]*"  | tr '\n' ' '; echo
  done
done
