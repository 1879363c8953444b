#!/bin/bash
# Round 6, last session: do non-temporal accesses help the E-sized fp32 streams?  The row GEMMs' result (written once, read by the NEXT
# kernel: 328 / 655 MB per launch) stored with `nt`; the segmented max / sums reading those rows with `nt` loads.  Variant libraries are
# built from sed-patched copies of the product sources (no switches in the product).
#   build here:  bash tools/experiments/nt_probe.sh build
#   GPU box:     bash tools/experiments/nt_probe.sh run > gpurun_out/r06zze_nt_probe.log 2>&1
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/neurips21-self-supervised-bug-detection-and-repair_amd/csrc
B=$R/tools/experiments/build
if [ "$1" = build ]; then
  mkdir -p $B
  (cd $C && make -s)
  FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I$C -I$R/include"
  # (a) nt stores of the row GEMM's result tile
  sed 's|if (mm < nrows \&\& n < N) \*reinterpret_cast<float4\*>(c + (size_t)(row0 + mm) \* ldc + n) = v;|if (mm < nrows \&\& n < N) __builtin_nontemporal_store(__builtin_bit_cast(ntf4, v), reinterpret_cast<ntf4*>(c + (size_t)(row0 + mm) * ldc + n));|; s|^typedef float f32x16 __attribute__((ext_vector_type(16)));|typedef float f32x16 __attribute__((ext_vector_type(16)));\ntypedef float ntf4 __attribute__((ext_vector_type(4)));|' $C/bl_gemm_h3.hip > $B/nt_gemm_h3.hip
  grep -c nontemporal $B/nt_gemm_h3.hip
  /opt/rocm/bin/hipcc $FLAGS -c $B/nt_gemm_h3.hip -o $B/nt_gemm_h3.o
  # (b) nt loads of the message / gradient rows in the per-node kernels
  sed 's|v\[u\]\[j\] = d < D ? row\[d\] : NEG_INF;|v[u][j] = d < D ? __builtin_nontemporal_load(\&row[d]) : NEG_INF;|; s|v\[u\]\[j\] = d < Din ? row\[d\] : 0.f;|v[u][j] = d < Din ? __builtin_nontemporal_load(\&row[d]) : 0.f;|' $C/bl_graph_ops.hip > $B/nt_graph_ops.hip
  grep -c nontemporal $B/nt_graph_ops.hip
  /opt/rocm/bin/hipcc $FLAGS -c $B/nt_graph_ops.hip -o $B/nt_graph_ops.o
  others=$(ls $C/build/*.o | grep -v "/bl_gemm_h3.o" | grep -v "/bl_graph_ops.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others $B/nt_gemm_h3.o $C/build/bl_graph_ops.o -o $B/libbuglab_hip_ntstore.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others $C/build/bl_gemm_h3.o $B/nt_graph_ops.o -o $B/libbuglab_hip_ntload.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others $B/nt_gemm_h3.o $B/nt_graph_ops.o -o $B/libbuglab_hip_ntboth.so
  rm -f $B/nt_*.o $B/nt_*.hip
  ls -la $B/libbuglab_hip_nt*.so
  exit 0
fi
cd $R
for v in product ntstore ntload ntboth; do
  if [ $v = product ]; then unset BL_HIP_LIB; else export BL_HIP_LIB=$B/libbuglab_hip_$v.so; fi
  echo "== $v"
  python tools/gemm_bench.py --which fwd_h3,nk_h3 2>/dev/null
  python tools/hbm_bench.py 2>/dev/null
  python bench.py --no-cpu-baseline --no-also --no-box 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels_serial']
print('bench', d['value'], 'graphs/s', d['ms_per_step'], 'ms;', {n: k[n]['ms_per_step'] for n in ('msg_dgrad_h3','msg_gemm_h3','segment_max_ln','node_grad_sums')})"
done
