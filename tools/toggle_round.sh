# Runs on the GPU box: the parity tests of the folded forward under the environment switches that select the
# older code paths, then one bench line each of configs 3 and 5.
set -e
cd $GRAFT_REPO_ROOT
TC_RES_STREAM=0 timeout -k 10 400 python -u -m pytest tests/test_gpu_lnfold.py tests/test_gpu_model.py -x -q --timeout 200 --timeout-method thread > gpurun_out/tog_nostream.log 2>&1
TC_HEAD_MFMA=0 timeout -k 10 400 python -u -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -x -q --timeout 200 --timeout-method thread -k "head or golden or trained or fold" > gpurun_out/tog_nohead.log 2>&1
timeout -k 10 300 python bench.py --config 3 --no-cpu-baseline > gpurun_out/b_cfg3.json 2> gpurun_out/b_cfg3.err
timeout -k 10 300 python bench.py --config 5 --no-cpu-baseline > gpurun_out/b_cfg5.json 2> gpurun_out/b_cfg5.err
echo done
