#!/bin/bash
# Run ON THE GPU BOX from the repo root:  bash tools/final_run.sh r03q
# GPU test-suite, smoke, the bench lines of every reported configuration and the rocprofv3 summaries (tools/collect_profiles.sh).
TAG=${1:?tag}
O=gpurun_out
mkdir -p $O
# (the bench lines first: run right behind the test-suite, the launch-bound seq-great entry of `also` once measured 766 instead of
# 3 200 sequences/s -- host contention from processes the tests leave winding down; r05m)
bash tools/collect_profiles.sh $TAG > $O/${TAG}_collect.log 2>&1
python bench.py --hidden 256 --graphs 32 --no-cpu-baseline --no-also > $O/${TAG}_bench_c3.json 2>/dev/null
python bench.py --hidden 256 --graphs 32 --degree powerlaw --no-cpu-baseline --no-also > $O/${TAG}_bench_c4.json 2>/dev/null
python bench.py --graphs 15 --no-cpu-baseline --no-also > $O/${TAG}_bench_30k.json 2>/dev/null
bash tools/seq_models_run.sh $TAG > $O/${TAG}_seq_models.log 2>&1
python bench.py --msg-gemm f16x1 --no-cpu-baseline --no-also > $O/${TAG}_bench_amp.json 2>/dev/null
python bench.py --aggregation sum --no-cpu-baseline --no-also > $O/${TAG}_bench_sum.json 2>/dev/null
python tools/gemm_bench.py --which fwd_x6,fwd_h3,nk_x6,nk_h3,wgrad_x6,wgrad_h3 --mixed 2 > $O/${TAG}_gemm_bench.log 2>&1
python tools/gemm_bench.py --which fwd_x6w,fwd_h3,nk_x6w,nk_h3,wgrad_x6,wgrad_h3 --din 256 --dm 256 --mixed 2 >> $O/${TAG}_gemm_bench.log 2>&1
python tools/hbm_bench.py > $O/${TAG}_hbm_bench.log 2>&1
timeout 600 python tools/dp_check.py --ranks 8 > $O/${TAG}_dp_check_8ranks_1gpu.log 2>&1; grep -c "step" $O/${TAG}_dp_check_8ranks_1gpu.log
timeout 600 python tools/e2e_train_bench.py --graphs 8192 --nodes 1300 --shards 128 --workers 32 --epochs 3 --real-validation 8 > $O/${TAG}_e2e_train_bench.log 2>&1; tail -3 $O/${TAG}_e2e_train_bench.log
timeout 2400 python -m pytest tests -x -q -m gpu --durations=10 > $O/${TAG}_gputest.log 2>&1; tail -2 $O/${TAG}_gputest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.log 2>&1; tail -2 $O/${TAG}_smoke.log
ls $O | grep $TAG | wc -l
