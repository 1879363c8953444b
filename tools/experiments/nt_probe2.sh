#!/bin/bash
# Round 6, last session, second probe: which cache-policy bits for the row GEMMs' streaming result stores, and do other once-written
# outputs want the hint too?  (nt_probe.sh found `nt` on the f16x3 row GEMM's result worth 20 % of the routed input gradient; the
# product has it since: csrc/bl_common.h::bl_store_streaming.)  Variant libraries from sed-patched copies of the product sources.
#   build here:  bash tools/experiments/nt_probe2.sh build
#   GPU box:     bash tools/experiments/nt_probe2.sh run > gpurun_out/r06zzf_nt_probe2.log 2>&1
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/neurips21-self-supervised-bug-detection-and-repair_amd/csrc
B=$R/tools/experiments/build
VARIANTS="plain sc1 sc0sc1 sc1nt sc0sc1nt x6nt segmaxnt sumsnt"
if [ "$1" = build ]; then
  mkdir -p $B/inc
  (cd $C && make -s)
  FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I$R/include"
  link() {  # name, replaced objects...
    local name=$1; shift
    local others=$(ls $C/build/*.o)
    for o in "$@"; do others=$(echo "$others" | grep -v "/$(basename $o)"); done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others "$@" -o $B/libbuglab_hip_$name.so
  }
  # cache-policy flavours of bl_store_streaming (the f16x3 row GEMM is its one user): a patched bl_common.h in front of the include path
  for v in plain sc1 sc0sc1 sc1nt sc0sc1nt; do
    case $v in plain) bits="";; sc1) bits=" sc1";; sc0sc1) bits=" sc0 sc1";; sc1nt) bits=" sc1 nt";; sc0sc1nt) bits=" sc0 sc1 nt";; esac
    rm -rf $B/inc_$v; mkdir -p $B/inc_$v
    python3 - "$C/bl_common.h" "$B/inc_$v/bl_common.h" "$bits" <<'PY'
import sys
src, dst, bits = sys.argv[1:4]
s = open(src).read()
old = "  __builtin_nontemporal_store(__builtin_bit_cast(bl_f32x4, v), reinterpret_cast<bl_f32x4*>(p));"
new = '  asm volatile("global_store_dwordx4 %0, %1, off' + bits + '" ::"v"(p), "v"(__builtin_bit_cast(bl_f32x4, v)) : "memory");'
assert old in s
open(dst, "w").write(s.replace(old, new).replace('#include "../../include/buglab_hip.h"', '#include "buglab_hip.h"'))
PY
    cp $C/bl_gemm_h3.hip $C/bl_x6_locate.h $C/bl_h3_image.h $B/inc_$v/
    /opt/rocm/bin/hipcc $FLAGS -I$B/inc_$v -c $B/inc_$v/bl_gemm_h3.hip -o $B/inc_$v/bl_gemm_h3.o
    link $v $B/inc_$v/bl_gemm_h3.o
  done
  # the bf16x6 row GEMMs' result (message GEMMs in bf16x6 mode, the dense node update, the sequence models' projections)
  mkdir -p $B/inc_x6nt
  sed 's|          \*reinterpret_cast<float4\*>(c + grow \* ldc + n) = v;|          bl_store_streaming(c + grow * ldc + n, v);|' $C/bl_gemm_x6.hip > $B/inc_x6nt/bl_gemm_x6.hip
  sed 's|if (mm_ < nrows \&\& n_ < N) \*reinterpret_cast<float4\*>(c + (size_t)(row0 + mm_) \* ldc + n_) = v_; |if (mm_ < nrows \&\& n_ < N) bl_store_streaming(c + (size_t)(row0 + mm_) * ldc + n_, v_); |' $C/bl_gemm_x6w.hip > $B/inc_x6nt/bl_gemm_x6w.hip
  grep -c bl_store_streaming $B/inc_x6nt/bl_gemm_x6.hip $B/inc_x6nt/bl_gemm_x6w.hip
  for f in bl_gemm_x6 bl_gemm_x6w; do /opt/rocm/bin/hipcc $FLAGS -I$C -c $B/inc_x6nt/$f.hip -o $B/inc_x6nt/$f.o; done
  link x6nt $B/inc_x6nt/bl_gemm_x6.o $B/inc_x6nt/bl_gemm_x6w.o
  # the segmented max's per-node outputs (aggregate, activation derivative, packed LayerNorm output)
  mkdir -p $B/inc_segmaxnt
  sed -e 's|      if (out) out\[(size_t)seg \* D + d\] = best\[j\];|      if (out) __builtin_nontemporal_store(best[j], \&out[(size_t)seg * D + d]);|' \
      -e 's|        dact\[(size_t)seg \* D + d\] = dv \* dscale;|        __builtin_nontemporal_store(dv * dscale, \&dact[(size_t)seg * D + d]);|' \
      -e '364s|o\[0\] = \(.*\);|__builtin_nontemporal_store((uint32_t)(\1), \&o[0]);|' \
      -e '365s|o\[halfD\] = \(.*\);|__builtin_nontemporal_store((uint32_t)(\1), \&o[halfD]);|' \
      -e '366s|o\[2 \* halfD\] = \(.*\);|__builtin_nontemporal_store((uint32_t)(\1), \&o[2 * halfD]);|' $C/bl_graph_ops.hip > $B/inc_segmaxnt/bl_graph_ops.hip
  grep -c nontemporal $B/inc_segmaxnt/bl_graph_ops.hip
  /opt/rocm/bin/hipcc $FLAGS -I$C -c $B/inc_segmaxnt/bl_graph_ops.hip -o $B/inc_segmaxnt/bl_graph_ops.o
  link segmaxnt $B/inc_segmaxnt/bl_graph_ops.o
  # the segmented sums: nt loads of the [E, 2 Din] rows, nt stores of the node gradient
  mkdir -p $B/inc_sumsnt
  sed -e 's|v\[u\]\[j\] = d < Din ? row\[d\] : 0.f;|v[u][j] = d < Din ? __builtin_nontemporal_load(\&row[d]) : 0.f;|' \
      -e 's|      if (d < split) g_h\[(size_t)n \* ld_gh + d\] = acc\[j\];|      if (d < split) __builtin_nontemporal_store(acc[j], \&g_h[(size_t)n * ld_gh + d]);|' $C/bl_graph_ops.hip > $B/inc_sumsnt/bl_graph_ops.hip
  grep -c nontemporal $B/inc_sumsnt/bl_graph_ops.hip
  /opt/rocm/bin/hipcc $FLAGS -I$C -c $B/inc_sumsnt/bl_graph_ops.hip -o $B/inc_sumsnt/bl_graph_ops.o
  link sumsnt $B/inc_sumsnt/bl_graph_ops.o
  rm -rf $B/inc_* $B/inc
  ls -la $B/ | grep -c "libbuglab_hip_"
  exit 0
fi
cd $R
show='
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["roofline"]["kernels_serial"]
print("bench", d["value"], d["unit"], d["ms_per_step"], "ms;", {n: v["ms_per_step"] for n, v in list(k.items())[:9]})'
for v in product $VARIANTS product; do
  if [ $v = product ]; then unset BL_HIP_LIB; else export BL_HIP_LIB=$B/libbuglab_hip_$v.so; fi
  echo "== $v"
  case $v in
    x6nt) python bench.py --no-cpu-baseline --no-also --no-box 2>/dev/null | python -c "$show"
          python bench.py --no-cpu-baseline --no-also --no-box --msg-gemm bf16x6 2>/dev/null | python -c "$show"
          python bench.py --no-cpu-baseline --no-also --no-box --model seq-great 2>/dev/null | python -c "$show"
          unset BL_HIP_LIB; echo "== product (bf16x6, seq-great)"
          python bench.py --no-cpu-baseline --no-also --no-box --msg-gemm bf16x6 2>/dev/null | python -c "$show"
          python bench.py --no-cpu-baseline --no-also --no-box --model seq-great 2>/dev/null | python -c "$show";;
    *) python tools/gemm_bench.py --which fwd_h3,nk_h3 2>/dev/null
       python bench.py --no-cpu-baseline --no-also --no-box 2>/dev/null | python -c "$show";;
  esac
done
