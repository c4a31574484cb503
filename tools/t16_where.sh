# Where k_ng_backward_t16's time goes: a diagnostic build (-DLG_NG_T16_DIAG) stamps the shader clock between the parts of a tile and a
# few waves print their totals (cycles per wave over its 10-11 tiles):  bash tools/t16_where.sh [extra -D flags]
R=$GRAFT_REPO_ROOT
cd $R
LIDARGS_EXTRA_HIPCC_FLAGS="-DLG_NG_T16_DIAG $*" python lidar-gs_amd/build_hip.py --force > /dev/null
timeout 120 python tools/time_decode.py 666667 6 3 hip 2>&1 | grep "^t16" | tail -8
python lidar-gs_amd/build_hip.py --force > /dev/null
