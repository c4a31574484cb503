# One workload's bench line + rocprofv3 kernel trace + PMC passes:  bash tools/profile_one.sh <prefix> <workload>
set -x
P=$1; wl=$2
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_LDS"
CMD="python $R/bench.py --workload $wl --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${P}_kt_$wl -o bench -- $CMD --steps 25 --warmup 5 > $R/gpurun_out/${P}_bench_${wl}_under_rocprof.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/${P}_fetch_$wl -o bench -- $CMD --steps 5 --warmup 2 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/${P}_write_$wl -o bench -- $CMD --steps 5 --warmup 2 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc $SQ -d $R/gpurun_out/${P}_sq_$wl -o bench -- $CMD --steps 5 --warmup 2 > $R/gpurun_out/${P}_sq_$wl.log 2>&1
cd $R
python tools/rocpd_stats.py gpurun_out/${P}_kt_$wl/bench_results.db > gpurun_out/${P}_bench_${wl}_kernel_stats.csv
python tools/pmc_traffic.py gpurun_out/${P}_fetch_$wl/bench_results.db gpurun_out/${P}_write_$wl/bench_results.db "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline" $wl > gpurun_out/${P}_pmc_traffic_$wl.json
python tools/pmc_sq.py $wl "rocprofv3 --kernel-trace --pmc $SQ -- python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline" gpurun_out/${P}_sq_$wl/bench_results.db > gpurun_out/${P}_pmc_sq_$wl.json
rm -rf gpurun_out/${P}_kt_$wl gpurun_out/${P}_fetch_$wl gpurun_out/${P}_write_$wl gpurun_out/${P}_sq_$wl
cp gpurun_out/${P}_pmc_traffic_$wl.json gpurun_out/${P}_pmc_sq_$wl.json profiles/ 2>/dev/null
timeout 600 python bench.py --workload $wl > gpurun_out/${P}_bench_${wl}.json 2> gpurun_out/${P}_bench_${wl}.err
