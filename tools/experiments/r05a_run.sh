#!/bin/bash
# GPU box: the fourth row-GEMM structure (8 waves, LDS-DMA, ping-pong; 256 x 128 tile or 128 x 256) against the shipped kernel
# at the LONG-K shapes (round 2 measured it only at K = 256): c2 plain + concat layer, c3 plain layer, c3 concat layer.
TAG=${1:-r05a}
O=gpurun_out; mkdir -p $O
cd tools/experiments
for lib in libx6v4.so libx6v4t.so; do
for cfg in "128000 640000 128 128" "128000 640000 256 256" "64000 320000 256 256" "64000 320000 512 512"; do
  set -- $cfg
  V4LIB=$lib NNODES=$1 NMSGS=$2 DIN=$3 DM=$4 timeout 300 python v4_bench.py 2>&1 | grep -v "^$\|amdgpu.ids"
done; done > ../../$O/${TAG}_v4_longk.log 2>&1
cat ../../$O/${TAG}_v4_longk.log
