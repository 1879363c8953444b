#!/bin/bash
# bench lines + kernel stats of the registry's other sequence encoders (seq-transformer, seq-gru) on the configs[4] workload
TAG=${1:?tag}
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for m in seq-transformer seq-gru; do
  python bench.py --model $m --no-cpu-baseline --no-also > $O/${TAG}_bench_${m}.json 2> $O/${TAG}_bench_${m}.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_${m} -o x -- python bench.py --model $m --steps 5 --warmup 2 --no-cpu-baseline --no-also > $O/${TAG}_rocprof_${m}.log 2>&1
  f=$(find $O/${TAG}_prof_${m} -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -25 "$f" > $O/${TAG}_bench_${m}_kernel_stats.csv
  rm -rf $O/${TAG}_prof_${m}
done
