# segment-plan A/B on the headline frame (edit the list per question):  bash tools/tune_plan.sh [workload]
cd $GRAFT_REPO_ROOT
WL=${1:-cfg3}
run() { echo "== $*"; env "$@" python bench.py --workload $WL --no-cpu-baseline --steps 150 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); s=j['stage_ms']; print(round(j['ms_per_step'],4), 'p1',s.get('render_pass1'),'p2',s.get('render_pass2'),'cmb',s.get('render_combine'),'bwd',s.get('render_bwd'))"; }
run A=1
run LIDARGS_ROUNDS=3
run LIDARGS_ROUNDS=4
run LIDARGS_ROUNDS=6
run LIDARGS_ROUNDS=8
run LIDARGS_ROUNDS=2,6
run LIDARGS_SEG_LEN=80 LIDARGS_ROUNDS=4
run LIDARGS_SEG_LEN=96 LIDARGS_ROUNDS=4
run LIDARGS_SEG_LEN=96 LIDARGS_ROUNDS=3
run LIDARGS_SEG_LEN=128 LIDARGS_ROUNDS=3 LIDARGS_HEAD=0 LIDARGS_P2_GROUP=1
run LIDARGS_SEG_LEN=128 LIDARGS_ROUNDS=2 LIDARGS_HEAD=0 LIDARGS_P2_GROUP=1
run LIDARGS_MAX_SEGMENTS=33
run LIDARGS_MAX_SEGMENTS=63
run A=1
