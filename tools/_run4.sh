O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu -k "not 64" > $O/r06d_gputest.log 2>&1; tail -5 $O/r06d_gputest.log
python bench.py --hidden 256 --graphs 32 --degree powerlaw --no-also --no-cpu-baseline > $O/r06d_bench_c4.json 2> $O/r06d_bench.err
python bench.py --hidden 256 --graphs 32 --no-also --no-cpu-baseline > $O/r06d_bench_c3.json 2>> $O/r06d_bench.err
python bench.py --no-also --no-cpu-baseline > $O/r06d_bench.json 2>> $O/r06d_bench.err
python -c "
import json
for f in ('bench','bench_c3','bench_c4'):
    j=json.loads(open('$O/r06d_%s.json'%f).read().strip().splitlines()[-1]); r=j['roofline']['kernels_serial']; print(f, j['value'], j['predict_graphs_per_s'], 'segmax', r['segment_max_ln'], 'sums', r['node_grad_sums'], j['box']['mfma_calib_tflops'], j['box']['sclk_mhz_step'], j['box']['gemm_calib_tflops'])"
