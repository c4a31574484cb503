for r in "5" "5,12" "4,10" "3,8,16" "6,14,24" "8"; do
  LIDARGS_ROUNDS=$r python bench.py --workload cfg3 --no-cpu-baseline --steps 40 | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stage_ms']; print('rounds $r', round(d['ms_per_step'],4), 'median', round(d['frame_ms_spread']['median'],4), 'pass1', s.get('render_pass1'), 'pass2', s.get('render_pass2'), 'bwd', s.get('render_bwd'))"
done
