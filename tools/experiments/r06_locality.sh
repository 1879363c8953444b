#!/bin/bash
# Round 6, last session: what would operand locality buy the f16x3 message GEMMs?  (The same probes were run on the bf16x6
# kernels in rounds 2 / 3, when the matrix pipes carried twice the work: all-in-L2 +15 %, (graph block, type) order slower.)
#   bash tools/experiments/r06_locality.sh > gpurun_out/r06zzc_locality.log 2>&1
W=fwd_h3,nk_h3,wgrad_h3
echo "== type-major, c2 layer shape (128 000 nodes)"
python tools/gemm_bench.py --which $W 2>/dev/null
echo "== every gathered row inside 4 000 nodes (2 MB of packed rows: L2-resident on every XCD)"
python tools/gemm_bench.py --which $W --nodes 4000 2>/dev/null
for c in 1 2 4 8; do
  echo "== (graph block, type) order, $c graphs per block"
  python tools/gemm_bench.py --which $W --order chunk --chunk $c 2>/dev/null
done
echo "== concat layer shape (Din 256, Dm 256): type-major / all-in-L2 / blocks of 2 and 4"
python tools/gemm_bench.py --which $W --din 256 --dm 256 2>/dev/null
python tools/gemm_bench.py --which $W --din 256 --dm 256 --nodes 4000 2>/dev/null
python tools/gemm_bench.py --which $W --din 256 --dm 256 --order chunk --chunk 2 2>/dev/null
python tools/gemm_bench.py --which $W --din 256 --dm 256 --order chunk --chunk 4 2>/dev/null
