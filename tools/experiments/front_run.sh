set -x
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_backbone.py -x -q -k "fused_front" > gpurun_out/r03e_front_test.log 2>&1; echo rc=$?; tail -15 gpurun_out/r03e_front_test.log
