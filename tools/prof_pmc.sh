#!/bin/bash
# Runs on the GPU box (via gpurun): one rocprofv3 counter pass per set, results under gpurun_out/pmc_<tag>/.
# usage: [BENCH_ARGS="--mode train"] tools/prof_pmc.sh <tag>
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd "$R"
run_pass () {
  name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$R/gpurun_out/pmc_${TAG}/$name" -- \
      python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > "$R/gpurun_out/pmc_${TAG}_$name.log" 2>&1 || return 1
}
run_pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU && \
run_pass fetch FETCH_SIZE && \
run_pass write WRITE_SIZE && \
run_pass grbm GRBM_GUI_ACTIVE
echo "pmc passes done rc=$?"
