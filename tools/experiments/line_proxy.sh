#!/bin/bash
# Timing proxy (results WRONG, memory pattern right): what if a k stage of a gathered f16x2 row were ONE 128-byte line?  The packed row is
# [plane][k group] -- a 32-k stage takes 64 bytes of each plane, i.e. two HALF lines per row and stage, the other halves one stage later.  An
# image laid out [stage][plane][k group] would make it one full line, requested once.  This variant only changes the ADDRESSES the row GEMM
# loads from (same bytes per lane, same LDS traffic, same MFMAs) to what that layout would give.
#   build here: bash tools/experiments/line_proxy.sh build ; GPU box: bash tools/experiments/line_proxy.sh run > gpurun_out/r06zzj_line_proxy.log 2>&1
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/neurips21-self-supervised-bug-detection-and-repair_amd/csrc
B=$R/tools/experiments/build
if [ "$1" = build ]; then
  (cd $C && make -s)
  mkdir -p $B/lp
  sed -e 's|const uint4\* src_ = base_ + (size_t)gr_ \* 2 \* (wj_ >> 3) + (kl_ >> 3); |const uint4* src_ = base_ + (size_t)gr_ * 2 * (wj_ >> 3) + ((kl_ >> 5) * 8 + p_kg); |' \
      -e 's|if (!ONE) ra\[i\]\[1\] = src_\[wj_ >> 3\]; |if (!ONE) ra[i][1] = src_[4]; |' $C/bl_gemm_h3.hip > $B/lp/bl_gemm_h3.hip
  diff $C/bl_gemm_h3.hip $B/lp/bl_gemm_h3.hip | grep -c "^>"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I$C -I$R/include -c $B/lp/bl_gemm_h3.hip -o $B/lp/bl_gemm_h3.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $C/build/*.o | grep -v /bl_gemm_h3.o) $B/lp/bl_gemm_h3.o -o $B/libbuglab_hip_lineproxy.so
  rm -rf $B/lp
  exit 0
fi
cd $R
for v in product lineproxy product lineproxy; do
  if [ $v = product ]; then unset BL_HIP_LIB; else export BL_HIP_LIB=$B/libbuglab_hip_$v.so; fi
  echo "== $v"
  python tools/gemm_bench.py --which fwd_h3,nk_h3 2>/dev/null
  python tools/gemm_bench.py --which fwd_h3,nk_h3 --din 256 --dm 256 2>/dev/null | grep -v "bit for bit"
done
