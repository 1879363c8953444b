#!/bin/bash
# Fourth probe of the `nt` store hint: seq-great's attention probabilities P and their gradient dS ([B H L, L] fp32: 268 MB each per layer,
# written once by the fused probabilities kernels, read by the products that follow).
#   build here:  bash tools/experiments/nt_probe4.sh build ;  GPU box:  bash tools/experiments/nt_probe4.sh run > gpurun_out/r06zzh_nt_probe4.log 2>&1
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/neurips21-self-supervised-bug-detection-and-repair_amd/csrc
B=$R/tools/experiments/build
if [ "$1" = build ]; then
  (cd $C && make -s)
  mkdir -p $B/p4
  sed -e 's|          \*reinterpret_cast<float4\*>(P + row \* L + key) = pr;|          bl_store_streaming(P + row * L + key, pr);|' \
      -e 's|if (key < L) \*reinterpret_cast<float4\*>(dS + row \* L + key) = \(make_float4(.*)\);|if (key < L) bl_store_streaming(dS + row * L + key, \1);|' $C/bl_seq_ops.hip > $B/p4/bl_seq_ops.hip
  echo "patched lines: $(grep -c bl_store_streaming $B/p4/bl_seq_ops.hip)"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I$C -I$R/include -c $B/p4/bl_seq_ops.hip -o $B/p4/bl_seq_ops.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $C/build/*.o | grep -v /bl_seq_ops.o) $B/p4/bl_seq_ops.o -o $B/libbuglab_hip_seqnt.so
  rm -rf $B/p4
  exit 0
fi
cd $R
show='
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["roofline"]["kernels_serial"]
print("bench", d["value"], d["unit"], d["ms_per_step"], "ms;", {n: v["ms_per_step"] for n, v in list(k.items())[:8]})'
for v in product seqnt product seqnt; do
  if [ $v = product ]; then unset BL_HIP_LIB; else export BL_HIP_LIB=$B/libbuglab_hip_$v.so; fi
  echo "== $v"
  python bench.py --no-cpu-baseline --no-also --no-box --model seq-great 2>/dev/null | python -c "$show"
done
