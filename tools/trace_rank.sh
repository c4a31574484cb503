# Kernel timeline of ONE virtual rank's frame of the sharded path (tools/time_shell.py replay: no RCCL time), with the gaps between launches.
#   bash tools/trace_rank.sh <out-prefix> [world] [cfg] [rank] [MODE]
P=${1:-x}; WORLD=${2:-8}; CFG=${3:-cfg3}; RANK=${4:-3}; export MODE=${5:-shell}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
FIRST=k_shell_flags; [ "$MODE" = wedge ] && FIRST=k_wedge_flags; [ "$LIDARGS_SELECT_FUSED" = 1 ] && FIRST=k_select_fused
ONLY_RANK=$RANK NOSTAGE=1 timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/${P}_kt -o t -- python $R/tools/time_shell.py $WORLD $CFG 100 > $R/gpurun_out/${P}_time.txt 2>&1
cd $R
DB=$(ls gpurun_out/${P}_kt/*/*.db gpurun_out/${P}_kt/*.db 2>/dev/null | head -1)
python tools/frame_timeline.py $DB $FIRST > gpurun_out/${P}_timeline.txt 2>&1
rm -rf gpurun_out/${P}_kt
grep "per-rank" gpurun_out/${P}_time.txt; tail -60 gpurun_out/${P}_timeline.txt
