# Where k_ng_backward_t16's time goes: a diagnostic build (-DLG_NG_T16_DIAG) stamps the shader clock between the parts of a tile (two
# waves of two workgroups print their totals) and can leave parts out (LIDARGS_NG_T16_SKIP, results meaningless):  bash tools/t16_where.sh
R=$GRAFT_REPO_ROOT
cd $R
LIDARGS_EXTRA_HIPCC_FLAGS=-DLG_NG_T16_DIAG python lidar-gs_amd/build_hip.py --force > /dev/null
timeout 120 python tools/time_decode.py 666667 6 3 hip 2>&1 | grep -v amdgpu.ids | tail -6
for s in 0 0 1 2 4 8 16 30 31; do
  echo -n "skip=$s: "
  LIDARGS_NG_T16_SKIP=$s timeout 120 python tools/time_decode.py 666667 6 20 hip 2>&1 | grep "anchor decode" | tail -1
done
python lidar-gs_amd/build_hip.py --force > /dev/null
