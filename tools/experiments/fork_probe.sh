#!/bin/bash
# When should a layer's routed weight gradient (side stream) start?  Product: together with the routed input gradient (both fork after the
# gradient operand is packed: two matrix-core kernels side by side).  Variants: after the input gradient has been LAUNCHED (next to the
# segmented sums and the next layer's node-update backward: bandwidth-bound neighbours), or after the sums.
# The variant sources are generated from csrc/bl_mp_layer.hip by tools/experiments/fork_variants.py (block moves only).
#   GPU box: bash tools/experiments/fork_probe.sh run > gpurun_out/r06zzn_fork_probe.log 2>&1
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/neurips21-self-supervised-bug-detection-and-repair_amd/csrc
B=$R/tools/experiments/build
if [ "$1" = build ]; then
  (cd $C && make -s)
  python3 $R/tools/experiments/fork_variants.py
  for v in forkafter_dgrad forkafter_sums; do
    mkdir -p $B/fp && cp $B/bl_mp_layer_$v.hip $B/fp/bl_mp_layer.hip
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I$C -I$R/include -c $B/fp/bl_mp_layer.hip -o $B/fp/m.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $C/build/*.o | grep -v /bl_mp_layer.o) $B/fp/m.o -o $B/libbuglab_hip_$v.so
    rm -rf $B/fp
  done
  exit 0
fi
cd $R
for v in product forkafter_dgrad forkafter_sums product forkafter_dgrad forkafter_sums; do
  if [ $v = product ]; then unset BL_HIP_LIB; else export BL_HIP_LIB=$B/libbuglab_hip_$v.so; fi
  echo "== $v"
  for cfg in "" "--hidden 256 --graphs 32" "--graphs 15"; do
    python bench.py --no-cpu-baseline --no-also --no-box --no-predict $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  $cfg', d['value'], d['unit'], d['ms_per_step'], 'ms')"
  done
done
