# segment-plan A/B on the headline frame at opacity scale 1 and 0.1:  bash tools/tune_plan_cfg3.sh
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" python bench.py --workload cfg3 --no-cpu-baseline --steps 150 $EXTRA 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); s=j['stage_ms']; print(round(j['ms_per_step'],4), 'blends', round(sum(s.get(k,0) for k in ('render_pass1','render_pass2','render_combine','render_bwd')),4), 'p1',s.get('render_pass1'),'p2',s.get('render_pass2'),'cmb',s.get('render_combine'),'bwd',s.get('render_bwd'))"; }
for EXTRA in "" "--opacity-scale 0.1"; do
echo "#### $EXTRA"
run A=1
for m in 29 37 45 53; do for r in 4 5 6; do run LIDARGS_MAX_SEGMENTS=$m LIDARGS_ROUNDS=$r; done; done
run LIDARGS_ROUNDS=5,12
run LIDARGS_ROUNDS=3,8
run A=1
done
