#!/bin/bash
O=gpurun_out; mkdir -p $O
python tools/node_bwd_bench.py > $O/r04i_node_bwd.log 2>&1; cat $O/r04i_node_bwd.log
timeout 600 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "node_update or forward_only" > $O/r04i_tests.log 2>&1; tail -40 $O/r04i_tests.log
