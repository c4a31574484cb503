# One measurement round on the GPU box: bench lines of every workload, rocprofv3 kernel trace + PMC passes of the raster workloads.
#   bash tools/profile_round.sh r02_a [quick]
# Writes gpurun_out/<P>_*; copy what should be judged into profiles/ (the PMC summaries are also copied there ON THE BOX, see below).
set -x
P=${1:-x}   # prefix of the files written under gpurun_out/
QUICK=${2:-}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
# counters first (bench.py looks its traffic / instruction counts up in the newest profiles/*_pmc_* of the workload), bench lines after
SQ="SQ_WAVES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_LDS"
for wl in cfg3 cfg2 cfg4 cfg5; do
  CMD="python $R/bench.py --workload $wl --no-cpu-baseline"
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${P}_kt_$wl -o bench -- $CMD --steps 25 --warmup 5 > $R/gpurun_out/${P}_bench_${wl}_under_rocprof.json 2>/dev/null
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/${P}_fetch_$wl -o bench -- $CMD --steps 5 --warmup 2 > /dev/null 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/${P}_write_$wl -o bench -- $CMD --steps 5 --warmup 2 > /dev/null 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc $SQ -d $R/gpurun_out/${P}_sq_$wl -o bench -- $CMD --steps 5 --warmup 2 > $R/gpurun_out/${P}_sq_$wl.log 2>&1
  ( cd $R
    python tools/rocpd_stats.py gpurun_out/${P}_kt_$wl/bench_results.db > gpurun_out/${P}_bench_${wl}_kernel_stats.csv
    python tools/pmc_traffic.py gpurun_out/${P}_fetch_$wl/bench_results.db gpurun_out/${P}_write_$wl/bench_results.db "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline" $wl > gpurun_out/${P}_pmc_traffic_$wl.json
    python tools/pmc_sq.py $wl "rocprofv3 --kernel-trace --pmc $SQ -- python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline" gpurun_out/${P}_sq_$wl/bench_results.db > gpurun_out/${P}_pmc_sq_$wl.json
    # the bench lines below price `traffic` and `compute` on the profile of THIS build: it has to be in profiles/ before they run
    cp gpurun_out/${P}_pmc_traffic_$wl.json gpurun_out/${P}_pmc_sq_$wl.json profiles/
    rm -rf gpurun_out/${P}_kt_$wl gpurun_out/${P}_fetch_$wl gpurun_out/${P}_write_$wl gpurun_out/${P}_sq_$wl )
  if [ -n "$QUICK" ]; then break; fi
done
cd /tmp
timeout 600 python $R/bench.py > $R/gpurun_out/${P}_bench.json 2> $R/gpurun_out/${P}_bench.err
for wl in cfg2 cfg4 cfg5 cfg1; do
  timeout 500 python $R/bench.py --workload $wl > $R/gpurun_out/${P}_bench_${wl}.json 2> $R/gpurun_out/${P}_bench_${wl}.err
done
timeout 300 python $R/bench.py --workload render_fps > $R/gpurun_out/${P}_bench_render_fps.json 2> $R/gpurun_out/${P}_bench_render_fps.err
if [ -z "$QUICK" ]; then
  timeout 300 python $R/bench.py --workload train_step --no-cpu-baseline > $R/gpurun_out/${P}_bench_train_step.json 2>/dev/null
  timeout 300 python $R/bench.py --workload decode > $R/gpurun_out/${P}_bench_decode.json 2>/dev/null
  timeout 300 python $R/bench.py --workload loss > $R/gpurun_out/${P}_bench_loss.json 2>/dev/null
fi
cd $R
tail -c 900 gpurun_out/${P}_bench.json
