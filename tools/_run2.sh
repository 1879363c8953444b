O=gpurun_out; mkdir -p $O
timeout 600 python tools/dp_check.py --ranks 8 > $O/r06b_dp_check_8ranks_1gpu.log 2>&1; tail -8 $O/r06b_dp_check_8ranks_1gpu.log
python bench.py > $O/r06b_bench.json 2> $O/r06b_bench.err; tail -c 400 $O/r06b_bench.err
python -c "
import json; j=json.loads(open('$O/r06b_bench.json').read().strip().splitlines()[-1]); print(j['value'], j['box']); print(j['roofline']['frac'], j['roofline'].get('frac_calibrated'))"
