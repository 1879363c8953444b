set -u
R=$(pwd); O=$R/gpurun_out; export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_seq
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_seq -o s -- python $R/bench.py --model seq-great --steps 6 --warmup 2 --no-cpu-baseline --no-predict --no-also --no-box > $O/r06s3_seq_rocprof.log 2>&1
f=$(find /tmp/prof_seq -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $O/r06s3_bench_seq_kernel_stats.csv
