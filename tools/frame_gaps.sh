# Kernel timeline of the headline frame with the gaps between launches:  bash tools/frame_gaps.sh <prefix> [workload] [extra bench args]
P=$1; wl=${2:-cfg3}; shift; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/${P}_kt -o t -- python $R/bench.py --workload $wl --no-cpu-baseline --steps 60 --warmup 10 "$@" > $R/gpurun_out/${P}_gaps_bench.json 2>/dev/null
cd $R
DB=$(ls gpurun_out/${P}_kt/*/*.db gpurun_out/${P}_kt/*.db 2>/dev/null | head -1)
python tools/frame_timeline.py $DB ${FIRST_KERNEL:-k_preprocess} > gpurun_out/${P}_frame_gaps_$wl.txt 2>&1
rm -rf gpurun_out/${P}_kt
cat gpurun_out/${P}_frame_gaps_$wl.txt | cut -c1-130
