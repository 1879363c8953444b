#!/bin/bash
# GPU box: graph-block-major work tables for the message GEMMs (tools/gemm_bench.py --sched)
O=gpurun_out; mkdir -p $O
python tools/gemm_bench.py --which fwd_x6,wgrad_x6 --kcaps 2816 --sched 1,2,4 > $O/r04f_sched_h128.log 2>&1
python tools/gemm_bench.py --din 256 --dm 256 --which fwd_x6,wgrad_x6 --kcaps 4096 --sched 1,2,4 > $O/r04f_sched_concat.log 2>&1
python tools/gemm_bench.py --which fwd_x6,wgrad_x6 --kcaps 2816 --sched 1,2 --sched-cap 1536 > $O/r04f_sched_h128_cap1536.log 2>&1
tail -n 30 $O/r04f_sched_h128.log $O/r04f_sched_concat.log $O/r04f_sched_h128_cap1536.log
