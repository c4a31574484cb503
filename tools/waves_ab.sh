# A/B of forced register budgets (waves per SIMD) for the blend kernels, built on the box:  bash tools/waves_ab.sh [surfel]
R=$GRAFT_REPO_ROOT; cd $R
run() { python bench.py --no-cpu-baseline $2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['stage_ms'].items() if 'render' in k})"; }
if [ "$1" = surfel ]; then
  for f in "" "-DLG_SF_FWD_WAVES=6" "-DLG_SF_BWD_WAVES=5" "-DLG_SF_BWD_WAVES=6" ""; do
    LIDARGS_EXTRA_HIPCC_FLAGS="$f" python lidar-gs_amd/build_hip.py --force > /dev/null 2>&1
    run "flags[$f] cfg5" "--workload cfg5"; run "flags[$f] cfg5 s=0.1" "--workload cfg5 --opacity-scale 0.1"
  done
else
  for f in "" "-DLG_BWD_WAVES=6" "-DLG_BWD_WAVES=8" "-DLG_FWD_WAVES=6" "-DLG_FWD_WAVES=8" ""; do
    LIDARGS_EXTRA_HIPCC_FLAGS="$f" python lidar-gs_amd/build_hip.py --force > /dev/null 2>&1
    run "flags[$f] s=1" ""; run "flags[$f] s=0.1" "--opacity-scale 0.1"
  done
fi
python lidar-gs_amd/build_hip.py --force > /dev/null 2>&1
