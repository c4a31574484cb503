for s in 1 0.1; do for h in 0 1; do
LIDARGS_HEAD=$h python bench.py --opacity-scale $s --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('s=$s HEAD=$h', round(d['value'],1), round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['stage_ms'].items() if 'render' in k})"
done; done
