#!/bin/bash
B=$PWD/tools/experiments/build
python tools/gemm_bench.py --which fwd_x6 2>&1 | grep fwd_x6
for v in 1 2 4 3 7; do
BL_HIP_LIB=$B/libbuglab_hip_abl$v.so python tools/gemm_bench.py --which fwd_x6 2>&1 | grep fwd_x6 | sed "s/^/  [ablate $v] /"
done
python tools/gemm_bench.py --which fwd_x6 2>&1 | grep fwd_x6
