set -x
python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "magnitude or f16x3" 2>&1 | tail -5 > gpurun_out/r06z2_magnitude.log
