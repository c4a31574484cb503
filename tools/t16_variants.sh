# A/B of compile-time variants of k_ng_backward_t16:  bash tools/t16_variants.sh "<flags of variant 1>" "<flags of variant 2>" ...
R=$GRAFT_REPO_ROOT
for fl in "$@"; do
  cd $R; LIDARGS_EXTRA_HIPCC_FLAGS="$fl" python lidar-gs_amd/build_hip.py --force > /dev/null
  cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/stg
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/stg -o t -- python $R/tools/time_decode.py 666667 6 40 hip > /tmp/stg.log 2>&1
  echo "flags [$fl]: $(grep 'anchor decode' /tmp/stg.log | tail -1 | sed 's/.*out; //')"; python $R/tools/rocpd_stats.py /tmp/stg/t_results.db | grep "k_ng_backward_t16\|k_ng_decode_t16\|k_ng_opacity_t16" | awk -F, '{printf "   %s: mean %.1f us\n", substr($0,1,42), $(NF-3)/1e3}'
done
cd $R; python lidar-gs_amd/build_hip.py --force > /dev/null
