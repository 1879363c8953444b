#!/bin/bash
V=$PWD/tools/experiments/build/libbuglab_hip_acc_major.so
for r in 1 2 3; do
python tools/gemm_bench.py --which fwd_x6,nk_x6 2>&1 | grep x6 | sed 's/^/[term-major] /'
BL_HIP_LIB=$V python tools/gemm_bench.py --which fwd_x6,nk_x6 2>&1 | grep x6 | sed 's/^/  [acc-major] /'
done
python tools/gemm_bench.py --din 256 --dm 256 --which fwd_x6,nk_x6 2>&1 | grep x6 | sed 's/^/[term-major] /'
BL_HIP_LIB=$V python tools/gemm_bench.py --din 256 --dm 256 --which fwd_x6,nk_x6 2>&1 | grep x6 | sed 's/^/  [acc-major] /'
