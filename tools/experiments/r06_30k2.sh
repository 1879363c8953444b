set -u
R=$(pwd); O=$R/gpurun_out
for g in 15 64; do
python bench.py --graphs $g --no-box --no-also --no-cpu-baseline --no-predict --steps 50 --warmup 10 > $O/r06t2_bench_g$g.json 2> $O/r06t2_bench_g$g.err
done
python bench.py --model seq-great --no-box --no-also --no-cpu-baseline --no-predict --steps 30 --warmup 5 > $O/r06t2_bench_seq.json 2> $O/r06t2_bench_seq.err
