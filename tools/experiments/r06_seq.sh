set -x
cd /root/repo
timeout 1500 python -m pytest tests/test_seq_great_gpu.py tests/test_cabi.py -x -q -m gpu 2>&1 | tail -30 > gpurun_out/r06s1_seqtests.log
