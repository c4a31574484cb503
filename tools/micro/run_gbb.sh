# bash tools/micro/run_gbb.sh "<variant numbers>"   -> one JSON line per variant of k_gaussian_backward (tools/micro/gauss_bwd_bench.cpp)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in ${1:-0}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics -DLG_GB_VARIANT=$v -w -I $R/include -I $R/lidar-gs_amd/csrc $R/tools/micro/gauss_bwd_bench.cpp \
        $R/lidar-gs_amd/csrc/preprocess.hip -o /tmp/gbb_$v 2>/dev/null || { echo "variant $v: build failed"; continue; }
  echo -n "variant $v: "; timeout 120 /tmp/gbb_$v
done
