set -x
P=${1:-x}   # prefix of the files written under gpurun_out/
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 500 python $R/bench.py > $R/gpurun_out/${P}_bench.json 2> $R/gpurun_out/${P}_bench.err
timeout 300 python $R/bench.py --workload cfg5 --no-cpu-baseline > $R/gpurun_out/${P}_bench_cfg5.json 2>/dev/null
timeout 300 python $R/bench.py --workload cfg4 --no-cpu-baseline > $R/gpurun_out/${P}_bench_cfg4.json 2>/dev/null
timeout 300 python $R/bench.py --workload train_step --no-cpu-baseline > $R/gpurun_out/${P}_bench_train_step.json 2>/dev/null
timeout 300 python $R/bench.py --workload decode > $R/gpurun_out/${P}_bench_decode.json 2>/dev/null
timeout 300 python $R/bench.py --workload loss > $R/gpurun_out/${P}_bench_loss.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${P}_kt -o bench -- python $R/bench.py --steps 25 --warmup 5 --no-cpu-baseline > $R/gpurun_out/${P}_bench_under_rocprof.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/${P}_fetch -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/${P}_write -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd $R
python tools/rocpd_stats.py gpurun_out/${P}_kt/bench_results.db > gpurun_out/${P}_kernel_stats.csv
python tools/pmc_traffic.py gpurun_out/${P}_fetch/bench_results.db gpurun_out/${P}_write/bench_results.db "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline" > gpurun_out/${P}_pmc_traffic.json
rm -rf gpurun_out/${P}_kt gpurun_out/${P}_fetch gpurun_out/${P}_write
tail -c 600 gpurun_out/${P}_bench.json
