#!/bin/bash
O=gpurun_out; mkdir -p $O
V=$PWD/tools/experiments/build/libbuglab_hip_pad13.so
timeout 600 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "gemm or bf16x6 or dense or routed or fused_layer" 2>&1 | tail -2
for r in 1 2 3; do
python tools/gemm_bench.py --which fwd_x6,nk_x6 2>&1 | grep x6 | sed 's/^/[swizzled] /'
BL_HIP_LIB=$V python tools/gemm_bench.py --which fwd_x6,nk_x6 2>&1 | grep x6 | sed 's/^/  [padded] /'
done
python tools/gemm_bench.py --din 256 --dm 256 --which fwd_x6,nk_x6 2>&1 | grep x6 | sed 's/^/[swizzled] /'
BL_HIP_LIB=$V python tools/gemm_bench.py --din 256 --dm 256 --which fwd_x6,nk_x6 2>&1 | grep x6 | sed 's/^/  [padded] /'
