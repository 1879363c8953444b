#!/bin/bash
# Build ablation variants of csrc/bl_node_bwd.hip as whole libraries under tools/experiments/build/ (they travel to the GPU box):
#   bash tools/experiments/node_bwd_variants.sh 1 2 4 8 16 32 63
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/neurips21-self-supervised-bug-detection-and-repair_amd/csrc
mkdir -p $R/tools/experiments/build
(cd $C && make -s)
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -DNB_ABLATE=$v -c $C/bl_node_bwd.hip -o $R/tools/experiments/build/nb_$v.o
  objs=$(ls $C/build/*.o | grep -v bl_node_bwd.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $R/tools/experiments/build/nb_$v.o -o $R/tools/experiments/build/libbuglab_hip_nb$v.so
  rm $R/tools/experiments/build/nb_$v.o
done
ls -la $R/tools/experiments/build/
