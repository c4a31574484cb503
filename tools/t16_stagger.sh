# k_ng_backward_t16's mean duration (kernel trace) against the start-up skew of its waves:  bash tools/t16_stagger.sh "0 1 2 3"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for st in $1; do
  rm -rf /tmp/stg
  LIDARGS_NG_T16_STAGGER=$st timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/stg -o t -- python $R/tools/time_decode.py 666667 6 40 hip > /dev/null 2>&1
  echo -n "stagger $st: "; python $R/tools/rocpd_stats.py /tmp/stg/t_results.db | grep "k_ng_backward_t16" | awk -F, '{printf "%s calls, mean %.1f us, min %.1f, max %.1f\n", $(NF-5), $(NF-3)/1e3, $(NF-1)/1e3, $NF/1e3}'
done
