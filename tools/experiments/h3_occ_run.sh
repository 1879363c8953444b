#!/bin/bash
# f16x3 kernels at four waves per SIMD (scratch spills and all): never run before (README: "56 - 104 B of scratch per lane: not run").
# Twin sources are generated from the product file; A/B on one box through BL_HIP_LIB.
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/neurips21-self-supervised-bug-detection-and-repair_amd/csrc
O=$R/gpurun_out
T=$R/tools/experiments
sed 's/__launch_bounds__(256, MASKED ? 2 : 3)/__launch_bounds__(256, 4)/' $C/bl_gemm_h3.hip > $T/bl_gemm_h3r_switches.hip
sed 's/__global__ __launch_bounds__(256, 2) void gemm_wgrad_h3_kernel/__global__ __launch_bounds__(256, 4) void gemm_wgrad_h3_kernel/' $C/bl_gemm_h3.hip > $T/bl_gemm_h3w_switches.hip
cp $T/bl_gemm_h3r_switches.hip $C/bl_gemm_h3r.hip; cp $T/bl_gemm_h3w_switches.hip $C/bl_gemm_h3w.hip
build() {  # name, source
  objs=$(ls $C/build/*.o | grep -v "/bl_gemm_h3.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I$C -I$R/include -c $2 -o $T/build/var_$1.o 2>&1 | tail -2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $T/build/var_$1.o -o $T/build/libbuglab_hip_$1.so
}
mkdir -p $T/build
if [ "$1" = build ]; then
build occ4rows $C/bl_gemm_h3r.hip
build occ4wgrad $C/bl_gemm_h3w.hip
fi
rm -f $C/bl_gemm_h3r.hip $C/bl_gemm_h3w.hip $T/bl_gemm_h3r_switches.hip $T/bl_gemm_h3w_switches.hip
[ "$1" = build ] && exit 0
for rep in 1 2; do
  for v in base occ4rows occ4wgrad; do
    if [ $v = base ]; then unset BL_HIP_LIB; else export BL_HIP_LIB=$T/build/libbuglab_hip_$v.so; fi
    python $R/bench.py --no-cpu-baseline --no-also --no-predict 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['roofline']['kernels_serial']
print('$v', d['value'], d['ms_per_step'], {n: k[n]['ms_per_step'] for n in ('msg_dgrad_h3', 'msg_gemm_h3', 'msg_wgrad_h3')})"
  done
done
