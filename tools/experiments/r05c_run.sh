#!/bin/bash
# GPU box: wide row GEMM (bl_gemm_rows_x6w) vs the shipped 128 x 128 kernel, forward and routed input gradient, long-K shapes
TAG=${1:-r05c}
O=gpurun_out; mkdir -p $O
for cfg in "128000 640000 256 256" "64000 320000 256 256" "64000 320000 512 512"; do
  set -- $cfg
  echo "== nodes $1 msgs $2 din $3 dm $4"
  timeout 600 python tools/gemm_bench.py --nodes $1 --msgs $2 --din $3 --dm $4 --which fwd_x6,fwd_x6w,nk_x6,nk_x6w --rounds 3 2>&1 | grep -v "amdgpu.ids"
done > $O/${TAG}_rows_wide.log 2>&1
cat $O/${TAG}_rows_wide.log
