#!/bin/bash
# GPU box: GPU test-suite, then A/B bench lines (fused node-update backward on / off)
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu > $O/r04j_gputest.log 2>&1; tail -5 $O/r04j_gputest.log
python bench.py --no-cpu-baseline --no-also > $O/r04j_bench.json 2> $O/r04j_bench.err
python bench.py --no-cpu-baseline --no-also --unfused-node-bwd > $O/r04j_bench_unfused.json 2> $O/r04j_bench_unfused.err
python - <<'PY'
import json
for n in ("r04j_bench", "r04j_bench_unfused"):
    try:
        d = json.load(open(f"gpurun_out/{n}.json"))
        ks = d["roofline"]["kernels_serial"]
        print(n, d["value"], d["ms_per_step"], "serial", d["roofline"]["serial_ms_per_step"], "predict", d["predict_graphs_per_s"])
        print("   ", {k: v["ms_per_step"] for k, v in ks.items()})
    except Exception as e:
        print(n, "failed", e); print(open(f"gpurun_out/{n}.err").read()[-1500:])
PY
