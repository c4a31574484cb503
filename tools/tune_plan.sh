cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" python bench.py --no-cpu-baseline --steps 150 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); s=j['stage_ms']; print(round(j['ms_per_step'],4), 'p1',s['render_pass1'],'p2',s['render_pass2'],'cmb',s['render_combine'],'bwd',s['render_bwd'])"; }
run A=1
run LIDARGS_ROUNDS=4
run LIDARGS_ROUNDS=6
run LIDARGS_ROUNDS=8
run LIDARGS_ROUNDS=3,8
run LIDARGS_SEG_LEN=96
run LIDARGS_MAX_SEGMENTS=63
run LIDARGS_MAX_SEGMENTS=33
run LIDARGS_SEG_LEN=96 LIDARGS_ROUNDS=4
run LIDARGS_RANGE_SORT_BITS=9
run LIDARGS_RANGE_SORT_BITS=10
