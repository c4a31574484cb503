#!/bin/bash
# Frame time and per-stage times of one workload (no CPU baseline leg):  bash tools/stage_ms.sh cfg5 [steps]
python bench.py --workload "${1:-cfg3}" --steps "${2:-200}" --no-cpu-baseline 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['value'], 1), round(d['ms_per_step'], 4), d['stage_ms'])"
