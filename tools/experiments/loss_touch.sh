#!/bin/bash
# The loss kernels with / without the up-front touch of the descriptor's index arrays (csrc/bl_loss.hip::warm_index_arrays): kernel times from
# rocprofv3 --kernel-trace --stats of short bench runs at 64 and 15 graphs; `lossold` = the library with the previous bl_loss.hip.
#   build here: bash tools/experiments/loss_touch.sh build ; GPU box: bash tools/experiments/loss_touch.sh > gpurun_out/r06zzq_loss_touch.log 2>&1
R=$(pwd)
if [ "$1" = build ]; then  # the library with bl_loss.hip as it was before commit 7cf7a0d
  C=$R/neurips21-self-supervised-bug-detection-and-repair_amd/csrc; B=$R/tools/experiments/build; mkdir -p $B/t
  (cd $C && make -s) && git -C $R show 7cf7a0d^:neurips21-self-supervised-bug-detection-and-repair_amd/csrc/bl_loss.hip > $B/t/bl_loss.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I$C -I$R/include -c $B/t/bl_loss.hip -o $B/t/l.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $C/build/*.o | grep -v /bl_loss.o) $B/t/l.o -o $B/libbuglab_hip_lossold.so
  rm -rf $B/t; exit 0
fi
cd /tmp && export TMPDIR=/tmp
for v in product lossold product lossold; do
  if [ $v = product ]; then unset BL_HIP_LIB; else export BL_HIP_LIB=$R/tools/experiments/build/libbuglab_hip_$v.so; fi
  for g in 64 15; do
    rm -rf /tmp/lt_prof
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lt_prof -o s -- python $R/bench.py --graphs $g --steps 10 --warmup 3 --no-cpu-baseline --no-also --no-box --no-predict > /tmp/lt_bench.json 2>/dev/null
    f=$(find /tmp/lt_prof -name '*kernel_stats.csv' | head -1)
    python - "$v" "$g" "$f" <<'PY'
import csv, json, sys
v, g, f = sys.argv[1:4]
rows = {r["Name"].split("(")[1 if r["Name"].startswith("(anonymous") else 0]: r for r in csv.DictReader(open(f))}
pick = lambda key: next((float(r["AverageNs"]) / 1e3 for n, r in rows.items() if key in n or key in r["Name"]), float("nan"))
d = json.loads(open("/tmp/lt_bench.json").read().strip().splitlines()[-1])
print(f"{v:8s} graphs {g:>3s}: bug_loss_fwd {pick('bug_loss_fwd_kernel'):6.1f} us  bug_loss_bwd {pick('bug_loss_bwd_kernel'):6.1f} us   {d['value']:8.1f} graphs/s {d['ms_per_step']:7.3f} ms (under the profiler)")
PY
  done
done
