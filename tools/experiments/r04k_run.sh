#!/bin/bash
O=gpurun_out; mkdir -p $O
V=$PWD/tools/experiments/build/libbuglab_hip_directstore.so
for r in 1 2; do
python tools/gemm_bench.py --which fwd_x6,nk_x6 > $O/r04k_staged_h128_$r.log 2>&1
BL_HIP_LIB=$V python tools/gemm_bench.py --which fwd_x6,nk_x6 > $O/r04k_direct_h128_$r.log 2>&1
done
python tools/gemm_bench.py --din 256 --dm 256 --which fwd_x6,nk_x6 > $O/r04k_staged_concat.log 2>&1
BL_HIP_LIB=$V python tools/gemm_bench.py --din 256 --dm 256 --which fwd_x6,nk_x6 > $O/r04k_direct_concat.log 2>&1
grep -H "x6" $O/r04k_*.log
timeout 900 python -m pytest tests -x -q -m gpu > $O/r04k_gputest.log 2>&1; tail -3 $O/r04k_gputest.log
python bench.py --no-cpu-baseline --no-also > $O/r04k_bench.json 2> $O/r04k_bench.err
BL_HIP_LIB=$V python bench.py --no-cpu-baseline --no-also > $O/r04k_bench_direct.json 2> $O/r04k_bench_direct.err
python - <<'PY'
import json
for n in ("r04k_bench", "r04k_bench_direct"):
    try:
        d = json.load(open(f"gpurun_out/{n}.json"))
        ks = d["roofline"]["kernels_serial"]
        print(n, d["value"], d["ms_per_step"], "serial", d["roofline"]["serial_ms_per_step"], "predict", d["predict_graphs_per_s"])
        print("   ", {k: v["ms_per_step"] for k, v in ks.items()})
    except Exception as e:
        print(n, "failed", e); print(open(f"gpurun_out/{n}.err").read()[-1500:])
PY
