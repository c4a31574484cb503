# Kernel trace of one workload's bench run (no PMC passes):  bash tools/kt.sh <prefix> <workload> [extra bench args]
P=$1; wl=$2; shift; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${P}_kt_$wl -o bench -- python $R/bench.py --workload $wl --no-cpu-baseline --steps 25 --warmup 5 "$@" > $R/gpurun_out/${P}_bench_${wl}_under_rocprof.json 2>/dev/null
cd $R
python tools/rocpd_stats.py gpurun_out/${P}_kt_$wl/bench_results.db > gpurun_out/${P}_bench_${wl}_kernel_stats.csv
rm -rf gpurun_out/${P}_kt_$wl
timeout 600 python bench.py --workload $wl --no-cpu-baseline "$@" > gpurun_out/${P}_bench_${wl}.json 2> gpurun_out/${P}_bench_${wl}.err
python - <<PY
import json,csv
d=json.loads(open("gpurun_out/${P}_bench_${wl}.json").read().strip().splitlines()[-1])
print("${wl}", round(d["value"],1), d["unit"], round(d["ms_per_step"],4), "ms", {k:v for k,v in d["config"].items() if "touched" in k or "taken" in k or "visible" in k})
rows=list(csv.DictReader(open("gpurun_out/${P}_bench_${wl}_kernel_stats.csv")))
for r in rows[:14]:
    print("%9.1f us x%-5s %s" % (float(r["AverageNs"])/1e3, r["Calls"], r["Name"][:90]))
PY
