# Frame time against opacity scale (the semi-transparent, early-training regime):  bash tools/thin_sweep.sh <prefix> [workloads...]
# Per (workload, scale): the bench line (un-profiled) and the kernel-trace means of the same command.
P=$1; shift
WLS=${@:-cfg3 cfg5}
R=$GRAFT_REPO_ROOT
for wl in $WLS; do
  for s in 1 0.3 0.1 0.03; do
    cd $R
    timeout 600 python bench.py --workload $wl --opacity-scale $s --no-cpu-baseline > gpurun_out/${P}_bench_${wl}_s$s.json 2> gpurun_out/${P}_bench_${wl}_s$s.err
    cd /tmp && export TMPDIR=/tmp
    timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${P}_kt -o bench -- python $R/bench.py --workload $wl --opacity-scale $s --no-cpu-baseline --steps 25 --warmup 5 > /dev/null 2>&1
    cd $R
    python tools/rocpd_stats.py gpurun_out/${P}_kt/bench_results.db > gpurun_out/${P}_bench_${wl}_s${s}_kernel_stats.csv
    rm -rf gpurun_out/${P}_kt
    python - <<PY
import json
d = json.load(open("gpurun_out/${P}_bench_${wl}_s$s.json"))
c = d["config"]
print("$wl s=$s: %.1f frames/s  %.3f ms  taken %s touched %s instances %s" % (d["value"], d["ms_per_step"], c.get("patch_instance_pairs_taken"), c.get("touched_gaussians", c.get("touched_surfels")), c.get("instances_binned")), {k: round(v, 3) for k, v in d.get("stage_ms", {}).items()})
PY
  done
done
