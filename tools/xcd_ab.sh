# A/B of an XCD-aware workgroup -> (patch, segment) mapping of the segmented blend launches (LG_XCD_CHUNK consecutive patches per XCD), built on the box:
#   bash tools/xcd_ab.sh
R=$GRAFT_REPO_ROOT; cd $R
run() { python bench.py --no-cpu-baseline $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print('$1', '$2', round(d['value'],1), round(d['ms_per_step'],4), 'p1', s.get('render_pass1'), 'p2', s.get('render_pass2'), 'bwd', s.get('render_bwd'))"; }
if [ -z "$1" ]; then
for f in "" "-DLG_XCD_CHUNK=1" "-DLG_XCD_CHUNK=4" "-DLG_XCD_CHUNK=16" "-DLG_XCD_CHUNK=64" ""; do
  LIDARGS_EXTRA_HIPCC_FLAGS="$f" python lidar-gs_amd/build_hip.py --force > /dev/null 2>&1
  run "[$f]" "--workload cfg3"; run "[$f]" "--workload cfg3 --opacity-scale 0.1"; run "[$f]" "--workload cfg5"; run "[$f]" "--workload cfg4"
done
python lidar-gs_amd/build_hip.py --force > /dev/null 2>&1
fi
# second question (bash tools/xcd_ab.sh order): the order the patches are dispatched in
if [ "$1" = order ]; then
for f in "" "-DLG_PATCH_ORDER=1" "-DLG_PATCH_ORDER=2" ""; do
  LIDARGS_EXTRA_HIPCC_FLAGS="$f" python lidar-gs_amd/build_hip.py --force > /dev/null 2>&1
  run "[$f]" "--workload cfg3"; run "[$f]" "--workload cfg3 --opacity-scale 0.1"; run "[$f]" "--workload cfg5"
done
python lidar-gs_amd/build_hip.py --force > /dev/null 2>&1
fi
