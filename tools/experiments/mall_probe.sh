#!/bin/bash
# Does the 256 MiB Infinity Cache carry the E-sized fp32 intermediates of a layer if the minibatch is processed graph block by
# graph block?  The per-node kernels and the message GEMMs at 4 ... 64 graphs (their buffers are re-used every iteration, so a
# working set below the cache size is served from it): rate of the algorithmic bytes per size.
O=gpurun_out
for g in 4 8 16 32 64; do
  echo "== graphs $g"
  python tools/hbm_bench.py --graphs $g 2>/dev/null
  python tools/gemm_bench.py --which fwd_h3,nk_h3,wgrad_h3 --nodes $((g*2000)) --msgs $((g*10000)) 2>/dev/null | tail -8
done
echo "== concat layer shape"
for g in 4 8 16 64; do
  echo "== graphs $g (Din 256 Dm 256)"
  python tools/hbm_bench.py --graphs $g --dm 256 --din 256 2>/dev/null
  python tools/gemm_bench.py --which fwd_h3,nk_h3,wgrad_h3 --din 256 --dm 256 --nodes $((g*2000)) --msgs $((g*10000)) 2>/dev/null | tail -8
done
