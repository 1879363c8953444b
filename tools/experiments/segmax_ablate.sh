#!/bin/bash
# Where does the segmented max + LayerNorm kernel's time go at the c2 layer shape (172 us per launch, 560 MB)?  Variant builds with one phase
# removed each (results wrong): the routing bitmask, the finish (activation, LayerNorm, five output arrays), the scan's row loads.
#   build here: bash tools/experiments/segmax_ablate.sh build ; GPU box: bash tools/experiments/segmax_ablate.sh run > gpurun_out/r06zzl_segmax_ablate.log 2>&1
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/neurips21-self-supervised-bug-detection-and-repair_amd/csrc
B=$R/tools/experiments/build
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I$C -I$R/include"
if [ "$1" = build ]; then
  (cd $C && make -s)
  mkdir -p $B/sa
  mk() {
    sed "$2" $C/bl_graph_ops.hip > $B/sa/bl_graph_ops.hip
    echo "$1: $(diff $C/bl_graph_ops.hip $B/sa/bl_graph_ops.hip | grep -c '^>') lines"
    /opt/rocm/bin/hipcc $FLAGS -c $B/sa/bl_graph_ops.hip -o $B/sa/g.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $C/build/*.o | grep -v /bl_graph_ops.o) $B/sa/g.o -o $B/libbuglab_hip_sa_$1.so
  }
  mk nobits 's|    if (winbits) segmax_winbits<NV>(seg_items, beg, end, D, barg, winbits);|    if (winbits \&\& D == 12345) segmax_winbits<NV>(seg_items, beg, end, D, barg, winbits);|'
  # finish removed: one conditional store keeps the scan alive
  mk nofinish 's|  segmax_finish<NV, HAS_LN, GELU2>(x, ldx, seg, D, act, best, barg, raw, werf, out, arg, ln_g, ln_b, eps, ln_out, mean_out, rstd_out, dact,|  if (best[0] == 12345.678f \&\& barg[0] == 77) out[seg] = best[0];\n  if (D == 12345) segmax_finish<NV, HAS_LN, GELU2>(x, ldx, seg, D, act, best, barg, raw, werf, out, arg, ln_g, ln_b, eps, ln_out, mean_out, rstd_out, dact,|'
  # scan without its row loads: the values come from the item ids
  mk noscan 's|          v\[u\]\[j\] = d < D ? row\[d\] : NEG_INF;|          v[u][j] = d < D ? (float)(e[u] + d) : NEG_INF;|'
  rm -rf $B/sa
  exit 0
fi
cd $R
show='
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["roofline"]["kernels_serial"]
print("bench", d["value"], d["unit"], d["ms_per_step"], "ms;", {n: k[n]["ms_per_step"] for n in ("segment_max_ln", "dense_fwd", "msg_gemm_h3")})'
for v in product sa_nobits sa_nofinish sa_noscan; do
  if [ $v = product ]; then unset BL_HIP_LIB; else export BL_HIP_LIB=$B/libbuglab_hip_$v.so; fi
  echo "== $v"
  python bench.py --no-cpu-baseline --no-also --no-box --no-predict 2>/dev/null | python -c "$show"
done
