#!/bin/bash
# Run ON THE GPU BOX from the repo root:  bash tools/pmc_passes.sh TAG "<command>" "<counters of pass 1>" "<counters of pass 2>" ...
# One rocprofv3 --pmc pass per counter group (own run each, --kernel-trace only: MI355X_MICROARCH.md "rocprofv3 PMC slots"),
# summarised per kernel into gpurun_out/TAG_pmc.json by tools/pmc_table.py.
set -u
TAG=$1; CMD=$2; shift 2
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
files=""
i=0
for grp in "$@"; do
  i=$((i+1))
  rm -rf /tmp/pmc_${TAG}_$i
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$i -o p -- $CMD > $O/${TAG}_pmc_$i.log 2>&1
  f=$(find /tmp/pmc_${TAG}_$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && files="$files $f" || { echo "pass $i ($grp): no counter file"; tail -5 $O/${TAG}_pmc_$i.log; }
done
cd $R
python tools/pmc_table.py $O/${TAG}_pmc.json $files
