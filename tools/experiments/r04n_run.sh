#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_edge_features.py tests/test_seq_great_gpu.py -x -q -m gpu > $O/r04n_tests.log 2>&1; tail -4 $O/r04n_tests.log
python bench.py > $O/r04n_bench.json 2> $O/r04n_bench.err
python bench.py --model seq-great --no-cpu-baseline > $O/r04n_bench_seq.json 2> $O/r04n_bench_seq.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04n_bench.json"))
print(d["value"], d["ms_per_step"], d["predict_graphs_per_s"], d["cpu_baseline"])
r = d["roofline"]; print({k: r[k] for k in r if k not in ("kernels_serial", "kernels_as_timed")})
for k, v in (d.get("also") or {}).items():
    print(k, v["value"], v["ms_per_step"], v["roofline"]["kernel"], v["roofline"]["frac"])
s = json.load(open("gpurun_out/r04n_bench_seq.json"))
print("seq", s["value"], s["ms_per_step"], s["roofline"]["kernel"], s["roofline"]["frac"])
PY
