#!/bin/bash
O=gpurun_out; mkdir -p $O
V=$PWD/tools/experiments/build/libbuglab_hip_masked3.so
for r in 1 2; do
python tools/gemm_bench.py --which nk_x6 2>&1 | grep nk_x6
BL_HIP_LIB=$V python tools/gemm_bench.py --which nk_x6 2>&1 | grep nk_x6 | sed 's/^/  [3 WGs] /'
done
python tools/gemm_bench.py --din 256 --dm 256 --which nk_x6 2>&1 | grep nk_x6
BL_HIP_LIB=$V python tools/gemm_bench.py --din 256 --dm 256 --which nk_x6 2>&1 | grep nk_x6 | sed 's/^/  [3 WGs] /'
