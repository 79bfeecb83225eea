set -e
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -k 10 300 python -u -m pytest tests/test_gpu_ops.py -x -q --timeout 200 --timeout-method thread -k "one_wave or large_tile" > gpurun_out/r3_p_tests.log 2>&1
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_p_bench.json 2> gpurun_out/r3_p_bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r3 -- python3 bench.py --streams 1 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3_p_bench_rocprof.json 2> gpurun_out/r3_p_rocprof.err
tools/prof_pmc.sh r3
BENCH_ARGS="--config 4" tools/prof_pmc.sh r3train
timeout -k 10 300 python bench.py --config 4 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3_p_bench_train.json 2> gpurun_out/r3_p_bench_train.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r3train -- python3 bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3_p_bench_train_rocprof.json 2> gpurun_out/r3_p_rocprof_train.err
echo done
