#!/bin/bash
TAG=${1:-r05k}
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/prof_seq
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_seq -o s -- python $R/bench.py --model seq-great --steps 6 --warmup 2 --no-cpu-baseline --no-predict --no-also > $O/${TAG}_seq_rocprof.log 2>&1
f=$(find /tmp/prof_seq -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/${TAG}_bench_seq_kernel_stats.csv
head -50 $O/${TAG}_bench_seq_kernel_stats.csv | cut -c1-160
