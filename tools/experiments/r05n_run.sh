#!/bin/bash
O=gpurun_out; mkdir -p $O
python -m cProfile -o /tmp/p.prof bench.py --no-cpu-baseline --no-predict --steps 6 --warmup 2 > $O/r05n_bench.json 2>$O/r05n_bench.err
python - <<PY > $O/r05n_pstats.txt 2>&1
import pstats
p = pstats.Stats("/tmp/p.prof")
p.sort_stats("tottime").print_stats(35)
p.sort_stats("cumulative").print_stats(45)
PY
python - <<PY
import json
j=json.loads(open("$O/r05n_bench.json").read().strip().splitlines()[-1])
print(j["value"], {k:(v["value"], v["ms_per_step"]) for k,v in j["also"].items()})
PY
head -60 $O/r05n_pstats.txt
