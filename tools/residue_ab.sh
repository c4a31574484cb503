# Does the open parity residue (tests/test_sweep_residue_gpu.py RESIDUE) follow the blend's arithmetic choices?  The listed sweep scenes with the
# library rebuilt on the box: default | library expf in the walks | no multiply-add contraction anywhere | both.     bash tools/residue_ab.sh [seed ...]
R=$GRAFT_REPO_ROOT; cd $R
SEEDS=${@:-976017 990343 976359}
for f in "" "-DLG_PRECISE_EXP" "-ffp-contract=off" "-DLG_PRECISE_EXP -ffp-contract=off"; do
  LIDARGS_EXTRA_HIPCC_FLAGS="$f" python lidar-gs_amd/build_hip.py --force > /dev/null 2>&1
  echo "=== flags [$f]"
  NO_ENVELOPE=1 python tools/thin_residue.py $SEEDS 2>/dev/null | cut -c1-260
done
python lidar-gs_amd/build_hip.py --force > /dev/null 2>&1
