# Kernel-trace means of the decode's backward launches, one launch (LIDARGS_NG_T16_PASSES=1) against two:  bash tools/t16_ab.sh
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for ps in 2 1 2 1; do
  rm -rf /tmp/stg
  LIDARGS_NG_T16_PASSES=$ps timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/stg -o t -- python $R/tools/time_decode.py 666667 6 40 hip > /tmp/stg.log 2>&1
  echo "passes $ps: $(grep 'anchor decode' /tmp/stg.log | tail -1 | sed 's/.*out; //')"; python $R/tools/rocpd_stats.py /tmp/stg/t_results.db | grep "k_ng_backward_t16\|k_ng_reduce_weight_grads(" | awk -F, '{n=$1; sub(/\(.*/,"",n); printf "   %s: %s calls, mean %.1f us\n", substr($0,1,40), $(NF-5), $(NF-3)/1e3}'
done
