#!/bin/bash
V=$PWD/tools/experiments/build/libbuglab_hip_plain2.so
for r in 1 2; do
python tools/gemm_bench.py --which fwd_x6 2>&1 | grep fwd_x6
BL_HIP_LIB=$V python tools/gemm_bench.py --which fwd_x6 2>&1 | grep fwd_x6 | sed 's/^/  [2 WGs] /'
done
python tools/gemm_bench.py --din 256 --dm 256 --which fwd_x6 2>&1 | grep fwd_x6
BL_HIP_LIB=$V python tools/gemm_bench.py --din 256 --dm 256 --which fwd_x6 2>&1 | grep fwd_x6 | sed 's/^/  [2 WGs] /'
