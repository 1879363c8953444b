O=gpurun_out; mkdir -p $O
timeout 600 python tools/dp_check.py --ranks 8 > $O/r06c_dp_check_8ranks_1gpu.log 2>&1; grep -v "^W09\|^\[W\|amdgpu.ids\|Gloo\|^$" $O/r06c_dp_check_8ranks_1gpu.log | tail -8
python bench.py > $O/r06c_bench.json 2> $O/r06c_bench.err; tail -c 400 $O/r06c_bench.err
python bench.py --graphs 15 --no-also --no-cpu-baseline > $O/r06c_bench_30k.json 2>> $O/r06c_bench.err
python bench.py --hidden 256 --graphs 32 --no-also --no-cpu-baseline > $O/r06c_bench_c3.json 2>> $O/r06c_bench.err
python bench.py --hidden 256 --graphs 32 --degree powerlaw --no-also --no-cpu-baseline > $O/r06c_bench_c4.json 2>> $O/r06c_bench.err
python -c "
import json
for f in ('bench','bench_30k','bench_c3','bench_c4'):
    j=json.loads(open('$O/r06c_%s.json'%f).read().strip().splitlines()[-1]); print(f, j['value'], j['box'])"
