#!/bin/bash
# GPU box: weight-gradient chunk cap A/B (tools/gemm_bench.py --kcaps, bench.py --wgrad-kcap)
O=gpurun_out; mkdir -p $O
python tools/gemm_bench.py --which wgrad_x6,wgrad_x6_t128 --kcaps 1024,2048,3072,4096,8192 > $O/r04e_kcap_h128.log 2>&1
python tools/gemm_bench.py --din 256 --dm 256 --which wgrad_x6,wgrad_x6_t128 --kcaps 1024,2048,4096,8192,16384 > $O/r04e_kcap_concat.log 2>&1
for c in 1024 2816 4096; do
  python bench.py --no-cpu-baseline --no-also --wgrad-kcap $c > $O/r04e_bench_kcap$c.json 2> $O/r04e_bench_kcap$c.err
done
tail -n 20 $O/r04e_kcap_h128.log $O/r04e_kcap_concat.log
python - <<'PY'
import json
for c in (1024, 2816, 4096):
    try:
        d = json.load(open(f"gpurun_out/r04e_bench_kcap{c}.json"))
        ks = d["roofline"]["kernels_serial"]
        print(c, d["value"], d["ms_per_step"], {k: ks[k]["ms_per_step"] for k in ("msg_wgrad_x6", "dense_wgrad")})
    except Exception as e:
        print(c, "failed", e)
PY
