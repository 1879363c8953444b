#!/bin/bash
TAG=${1:-r05g}
O=gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu > $O/${TAG}_gputest.log 2>&1; tail -3 $O/${TAG}_gputest.log
python bench.py --hidden 256 --graphs 32 --no-cpu-baseline --no-also > $O/${TAG}_bench_c3.json 2>$O/${TAG}_bench_c3.err; python - <<PY
import json
for f in ("$O/${TAG}_bench_c3.json",):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, j["value"], j["ms_per_step"], {k:v["ms_per_step"] for k,v in j["roofline"]["kernels_serial"].items()})
    except Exception as e: print(f, "ERR", e)
PY
python bench.py --no-cpu-baseline --no-also > $O/${TAG}_bench.json 2>$O/${TAG}_bench.err; python - <<PY
import json
for f in ("$O/${TAG}_bench.json",):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, j["value"], j["ms_per_step"], {k:v["ms_per_step"] for k,v in j["roofline"]["kernels_serial"].items()})
    except Exception as e: print(f, "ERR", e)
PY
