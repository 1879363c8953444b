O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_kernels.py tests/test_train_cli_gpu.py -x -q -m gpu -k "trajector or f16x3 or train" -s > $O/r06l_tests.log 2>&1; tail -6 $O/r06l_tests.log
