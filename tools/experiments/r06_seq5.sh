set -u
R=$(pwd); O=$R/gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_seq_great_gpu.py tests/test_cabi.py -x -q -m gpu 2>&1 | tail -4 > $O/r06s7_seqtests.log
python bench.py --model seq-great --no-box > $O/r06s7_bench_seq.json 2> $O/r06s7_bench_seq.err
cd /tmp
rm -rf /tmp/prof_seq
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_seq -o s -- python $R/bench.py --model seq-great --steps 6 --warmup 2 --no-cpu-baseline --no-predict --no-also --no-box > $O/r06s7_seq_rocprof.log 2>&1
f=$(find /tmp/prof_seq -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $O/r06s7_bench_seq_kernel_stats.csv
