#!/bin/bash
# row GEMM variants timed back to back (power-capped) and alternating with an HBM-bound filler (the training step's conditions)
O=gpurun_out; mkdir -p $O
for cfg in "128000 640000 128 128" "128000 640000 256 256" "64000 320000 256 256" "64000 320000 512 512"; do
  set -- $cfg
  for mixed in 0 2; do
    echo "== nodes $1 msgs $2 din $3 dm $4 mixed $mixed"
    timeout 600 python tools/gemm_bench.py --nodes $1 --msgs $2 --din $3 --dm $4 --which fwd_x6,fwd_x6w,nk_x6,nk_x6w,wgrad_x6 --rounds 3 --iters 20 --mixed $mixed 2>&1 | grep -v "amdgpu.ids\|bit for bit\|skipping"
  done
done > $O/r05t_mixed.log 2>&1
cat $O/r05t_mixed.log
