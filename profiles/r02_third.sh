cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02c
timeout -k 10 1100 python -m pytest tests/test_multi_sim.py tests/test_multi_gpu_gloo.py tests/test_train_script.py -m gpu -q > gpurun_out/r02c/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -40 gpurun_out/r02c/pytest_gpu.log
