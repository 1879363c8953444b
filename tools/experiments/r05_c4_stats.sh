#!/bin/bash
# GPU box: per-kernel times of the power-law configuration (c4) next to c3: where do the per-node kernels lose?
R=$(pwd); O=$R/gpurun_out; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
for cfg in c3 c4; do
  extra=""; [ $cfg = c4 ] && extra="--degree powerlaw"
  rm -rf /tmp/prof_$cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$cfg -o s -- python $R/bench.py --hidden 256 --graphs 32 $extra --serial --steps 6 --warmup 2 --no-cpu-baseline --no-predict --no-also > /dev/null 2>&1
  f=$(find /tmp/prof_$cfg -name '*kernel_stats.csv' | head -1); cp $f $O/r05_${cfg}_serial_kernel_stats.csv
done
python - <<PY
import csv
for cfg in ("c3", "c4"):
    rows = list(csv.DictReader(open("$O/r05_%s_serial_kernel_stats.csv" % cfg)))
    steps = next(int(x["Calls"]) for x in rows if x["Name"].startswith("adam_clip_kernel"))
    print("==", cfg, "kernel ms per step", sum(int(x["TotalDurationNs"]) for x in rows) / steps / 1e6)
    for x in rows:
        if any(s in x["Name"] for s in ("segment_max", "mp_scatter", "node_bwd", "routed_dgrad")):
            print(f'{x["Name"][:70]:70s} calls/step {int(x["Calls"]) / steps:5.1f} ms/step {int(x["TotalDurationNs"]) / steps / 1e6:7.3f} avg us {float(x["AverageNs"]) / 1e3:8.1f} max us {float(x["MaxNs"]) / 1e3:8.1f}')
PY
