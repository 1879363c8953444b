#!/bin/bash
# An experiment build of the library with one source compiled differently (it travels to the GPU box with the tree):
#   bash tools/experiments/variant_lib.sh NAME bl_gemm_x6.hip -DX6_STAGED_STORE=0   -> tools/experiments/build/libbuglab_hip_NAME.so
# Use it with BL_HIP_LIB=$PWD/tools/experiments/build/libbuglab_hip_NAME.so (hip_ops.LIB_PATH).
set -e
NAME=$1; SRC=$2; shift 2
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/neurips21-self-supervised-bug-detection-and-repair_amd/csrc
B=$R/tools/experiments/build
mkdir -p $B
(cd $C && make -s)
# a source whose measurement switches were moved out of the product has a twin here (bl_gemm_x6.hip -> bl_gemm_x6_switches.hip)
TWIN=$R/tools/experiments/${SRC%.hip}_switches.hip
IN=$C/$SRC; [ -f "$TWIN" ] && IN=$TWIN
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I$C -I$R/include "$@" -c $IN -o $B/var_$NAME.o
objs=$(ls $C/build/*.o | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $B/var_$NAME.o -o $B/libbuglab_hip_$NAME.so
rm $B/var_$NAME.o
ls -la $B/libbuglab_hip_$NAME.so
