#!/bin/bash
# Fifth probe of the `nt` hint: the node-level streams around the node update's backward kernel -- its inputs g_out / h_out (each line read
# exactly once), its outputs (the fp32 node gradient gq that the packer reads next, the packed g_z that the dense weight gradient reads) -- and
# the packers' fp32 input.
#   build here: bash tools/experiments/nt_probe5.sh build ; GPU box: bash tools/experiments/nt_probe5.sh run > gpurun_out/r06zzk_nt_probe5.log 2>&1
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/neurips21-self-supervised-bug-detection-and-repair_amd/csrc
B=$R/tools/experiments/build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I$C -I$R/include"
if [ "$1" = build ]; then
  (cd $C && make -s)
  mkdir -p $B/p5
  NTF4='__builtin_bit_cast(float4, __builtin_nontemporal_load(reinterpret_cast<const bl_f32x4*>(\1)))'
  # (a) inputs: nt loads
  sed -e "s|rg0 = \*reinterpret_cast<const float4\*>(\(gp_\)); |rg0 = $NTF4; |" -e "s|rg1 = \*reinterpret_cast<const float4\*>(\(gp_ + 4\)); |rg1 = $NTF4; |" \
      -e "s|ry0 = \*reinterpret_cast<const float4\*>(\(yp_\)); |ry0 = $NTF4; |" -e "s|ry1 = \*reinterpret_cast<const float4\*>(\(yp_ + 4\)); |ry1 = $NTF4; |" $C/bl_node_bwd.hip > $B/p5/in.hip
  # (b) outputs: nt stores
  sed -e 's|if (gq_f32) \*reinterpret_cast<float4\*>(gq_f32 + gr \* Dm + col) = v;|if (gq_f32) bl_store_streaming(gq_f32 + gr * Dm + col, v);|' \
      -e 's|      o_\[0\] = ph_;  |      __builtin_nontemporal_store(__builtin_bit_cast(bl_f32x4, ph_), reinterpret_cast<bl_f32x4*>(o_));  |' \
      -e 's|      o_\[kq\] = pm_;  |      __builtin_nontemporal_store(__builtin_bit_cast(bl_f32x4, pm_), reinterpret_cast<bl_f32x4*>(o_ + kq));  |' \
      -e 's|      o_\[2 \* kq\] = pl_;  |      __builtin_nontemporal_store(__builtin_bit_cast(bl_f32x4, pl_), reinterpret_cast<bl_f32x4*>(o_ + 2 * kq));  |' $C/bl_node_bwd.hip > $B/p5/out.hip
  sed -e "s|rg0 = \*reinterpret_cast<const float4\*>(\(gp_\)); |rg0 = $NTF4; |" -e "s|rg1 = \*reinterpret_cast<const float4\*>(\(gp_ + 4\)); |rg1 = $NTF4; |" \
      -e "s|ry0 = \*reinterpret_cast<const float4\*>(\(yp_\)); |ry0 = $NTF4; |" -e "s|ry1 = \*reinterpret_cast<const float4\*>(\(yp_ + 4\)); |ry1 = $NTF4; |" $B/p5/out.hip > $B/p5/both.hip
  for v in in out both; do
    echo "$v: $(diff $C/bl_node_bwd.hip $B/p5/$v.hip | grep -c '^>') lines"
    cp $B/p5/$v.hip $B/p5/bl_node_bwd.hip
    /opt/rocm/bin/hipcc $FLAGS -c $B/p5/bl_node_bwd.hip -o $B/p5/nb_$v.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $C/build/*.o | grep -v /bl_node_bwd.o) $B/p5/nb_$v.o -o $B/libbuglab_hip_nb$v.so
  done
  # (c) the f16x2 packer's fp32 input
  sed -e "s|const float4 a = \*reinterpret_cast<const float4\*>(\(x + r \* ld + 8 \* kg\));|const float4 a = $NTF4;|" \
      -e "s|const float4 b = \*reinterpret_cast<const float4\*>(\(x + r \* ld + 8 \* kg + 4\));|const float4 b = $NTF4;|" $C/bl_gemm_h3.hip > $B/p5/bl_gemm_h3.hip
  echo "pack: $(diff $C/bl_gemm_h3.hip $B/p5/bl_gemm_h3.hip | grep -c '^>') lines"
  /opt/rocm/bin/hipcc $FLAGS -c $B/p5/bl_gemm_h3.hip -o $B/p5/h3.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $C/build/*.o | grep -v /bl_gemm_h3.o) $B/p5/h3.o -o $B/libbuglab_hip_packin.so
  rm -rf $B/p5
  exit 0
fi
cd $R
show='
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["roofline"]["kernels_serial"]
print("bench", d["value"], d["unit"], d["ms_per_step"], "ms;", {n: k[n]["ms_per_step"] for n in ("node_update_bwd", "pack_rows", "pack_gq_h3", "dense_wgrad", "msg_dgrad_h3", "msg_gemm_h3")})'
for v in product nbin nbout nbboth packin product; do
  if [ $v = product ]; then unset BL_HIP_LIB; else export BL_HIP_LIB=$B/libbuglab_hip_$v.so; fi
  echo "== $v"
  python bench.py --no-cpu-baseline --no-also --no-box 2>/dev/null | python -c "$show"
done
