# Runs on the GPU box: kernel traces of one bench command with several builds of the library on the same box
# (TC_LIB_PATH selects the build), each build twice, for per-kernel comparisons that a whole-step time cannot
# resolve (repeats agree to 0.1 %).
#   BENCH_ARGS="--config 4 --steps 5 --warmup 2" tools/trace_ab.sh tag libA.so libB.so [libC.so]
# An argument may carry environment switches after the path: lib.so,TC_RES_STREAM=0,TC_GEMM_W=1
set -e
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tag=$1; shift
for rep in 1 2; do
  i=0
  for lib in "$@"; do
    i=$((i+1))
    IFS=, read -r path sw <<< "$lib"
    export TC_LIB_PATH=$GRAFT_REPO_ROOT/$path
    ( for kv in ${sw//,/ }; do export "$kv"; done
      timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/trace_${tag}_${i}_$rep -- python3 bench.py ${BENCH_ARGS:-} --no-cpu-baseline > gpurun_out/trace_${tag}_${i}_$rep.json 2> gpurun_out/trace_${tag}_${i}_$rep.err )
    python3 tools/rocpd_stats.py /tmp/trace_${tag}_${i}_$rep/*/*_results.db gpurun_out/trace_${tag}_${i}_$rep.csv > /dev/null
    rm -rf /tmp/trace_${tag}_${i}_$rep
  done
done
echo done
