set -u
R=$(pwd); O=$R/gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_seq_great_gpu.py tests/test_cabi.py -x -q -m gpu 2>&1 | tail -4 > $O/r06s4_seqtests.log
python bench.py --model seq-great --no-box > $O/r06s4_bench_seq.json 2> $O/r06s4_bench_seq.err
BL_SIDE_STREAM=0 python bench.py --model seq-great --no-box > $O/r06s4_bench_seq_noside.json 2> $O/r06s4_bench_seq_noside.err
