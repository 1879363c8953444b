#!/bin/bash
# Third probe of the cache-policy hint: `nt` LOADS of the gathered operand rows of the f16x3 message GEMMs (each row is fetched once per
# message that uses it, ~10 times per layer, but far apart in time: no L2 reuse to lose, and the weights / routing bits would keep theirs).
#   build here:  bash tools/experiments/nt_probe3.sh build
#   GPU box:     bash tools/experiments/nt_probe3.sh run > gpurun_out/r06zzg_nt_probe3.log 2>&1
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/neurips21-self-supervised-bug-detection-and-repair_amd/csrc
B=$R/tools/experiments/build
VARIANTS="fwdA wgradA wgradAG all"
if [ "$1" = build ]; then
  (cd $C && make -s)
  FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I$C -I$R/include"
  NTL='__builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const ntu4*>(\&\1)))'
  TYPEDEF='s|^typedef float f32x16 __attribute__((ext_vector_type(16)));|typedef float f32x16 __attribute__((ext_vector_type(16)));\ntypedef unsigned ntu4 __attribute__((ext_vector_type(4)));|'
  # forward (unmasked) row GEMM: the gathered rows of h
  FWD_A="s|ra\[i\]\[0\] = \(src_\[0\]\);  |ra[i][0] = MASKED ? src_[0] : $NTL;  |; s|if (!ONE) ra\[i\]\[1\] = \(src_\[wj_ >> 3\]\);  |if (!ONE) ra[i][1] = MASKED ? src_[wj_ >> 3] : $NTL;  |"
  WG_A="s|ra\[i\]\[0\] = \(a_\[0\]\);  |ra[i][0] = $NTL;  |; s|if (!ONE) ra\[i\]\[1\] = \(a_\[awg\]\);  |if (!ONE) ra[i][1] = $NTL;  |"
  WG_G="s|rb\[i\]\[0\] = \(g_\[0\]\);  |rb[i][0] = $NTL;  |; s|if (!ONE) rb\[i\]\[1\] = \(g_\[gwg\]\);  |if (!ONE) rb[i][1] = $NTL;  |"
  mk() {  # name, sed program
    mkdir -p $B/p3_$1
    sed -e "$TYPEDEF" -e "$2" $C/bl_gemm_h3.hip > $B/p3_$1/bl_gemm_h3.hip
    echo "$1: $(grep -c nontemporal_load $B/p3_$1/bl_gemm_h3.hip) patched lines"
    /opt/rocm/bin/hipcc $FLAGS -c $B/p3_$1/bl_gemm_h3.hip -o $B/p3_$1/bl_gemm_h3.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $C/build/*.o | grep -v /bl_gemm_h3.o) $B/p3_$1/bl_gemm_h3.o -o $B/libbuglab_hip_$1.so
    rm -rf $B/p3_$1
  }
  mk fwdA "$FWD_A"
  mk wgradA "$WG_A"
  mk wgradAG "$WG_A; $WG_G"
  mk all "$FWD_A; $WG_A; $WG_G"
  exit 0
fi
cd $R
show='
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["roofline"]["kernels_serial"]
print("bench", d["value"], d["unit"], d["ms_per_step"], "ms;", {n: v["ms_per_step"] for n, v in list(k.items())[:5]})'
for v in product $VARIANTS product; do
  if [ $v = product ]; then unset BL_HIP_LIB; else export BL_HIP_LIB=$B/libbuglab_hip_$v.so; fi
  echo "== $v"
  python tools/gemm_bench.py --which fwd_h3,nk_h3,wgrad_h3 2>/dev/null
  python bench.py --no-cpu-baseline --no-also --no-box 2>/dev/null | python -c "$show"
done
