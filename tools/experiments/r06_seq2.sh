set -x
cd /root/repo
python bench.py --model seq-great --no-box > gpurun_out/r06s2_bench_seq.json 2> gpurun_out/r06s2_bench_seq.err
BL_FUSED_GREAT_LAYER=0 python bench.py --model seq-great --no-box > gpurun_out/r06s2_bench_seq_opbyop.json 2> gpurun_out/r06s2_bench_seq_opbyop.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_seq -o seq -- python /root/repo/bench.py --model seq-great --no-box --steps 10 > /dev/null 2>&1
cp /tmp/prof_seq/*kernel_stats.csv /root/repo/gpurun_out/r06s2_bench_seq_kernel_stats.csv 2>/dev/null || find /tmp/prof_seq -name "*kernel_stats.csv" -exec cp {} /root/repo/gpurun_out/r06s2_bench_seq_kernel_stats.csv \;
