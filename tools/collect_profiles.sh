#!/bin/bash
# Run ON THE GPU BOX from the repo root:  bash tools/collect_profiles.sh r02f
# Writes raw outputs under gpurun_out/; `python tools/summarize_profiles.py r02f` (build container) turns them
# into the committed summaries under profiles/.
set -u
TAG=${1:?tag}
R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
PROF="--steps 6 --warmup 2 --no-cpu-baseline --no-predict --no-also --no-box"
if [ -z "${SKIP_BENCH:-}" ]; then
python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
python bench.py --model seq-great > $O/${TAG}_bench_seq.json 2> $O/${TAG}_bench_seq.err
fi
cd /tmp
stats() {  # name, env, bench args
  rm -rf /tmp/prof_$1
  env $2 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1 -o s -- python $R/bench.py $3 > $O/${TAG}_$1_rocprof.log 2>&1
  f=$(find /tmp/prof_$1 -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $O/${TAG}_$1_kernel_stats.csv
}
stats bench "X=1" "$PROF"
stats bench_serial "X=1" "--serial $PROF"
stats bench_seq "X=1" "--model seq-great $PROF"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$C -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-predict --no-also --no-box > $O/${TAG}_pmc_$C.log 2>&1
  f=$(find /tmp/pmc_$C -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python $R/tools/pmc_sum.py $f $C > $O/${TAG}_pmc_$C.json
done
# SQ counters of the serial run (own passes: MI355X_MICROARCH.md "rocprofv3 PMC slots"): matrix-pipe busy cycles against the
# kernel's cycle count, L2 hits / misses -> gpurun_out/${TAG}_sq_pmc.json (tools/pmc_table.py)
cd $R
bash tools/pmc_passes.sh ${TAG}_sq "python $R/bench.py --serial --steps 2 --warmup 1 --no-cpu-baseline --no-predict --no-also --no-box" \
  "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" > $O/${TAG}_sq_passes.log 2>&1
ls -la $O | grep $TAG
