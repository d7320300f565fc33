#!/bin/bash
# GPU box: one PMC pass of the default form on 03 (1080p / 256 spp) and 04 (4K / 256 spp) -> gpurun_out/pmc_one/   (tools/pmc_one.sh [tag])
set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_one${1:+_$1}; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for which in 03 04; do
  D=$OUT/$which; mkdir -p $D
  timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $D -o p -- python3 $R/tools/sorted_ab.py time$which 0 > $D/run.log 2>&1 || exit 1
  echo "== $which"; tail -1 $D/run.log; python3 $R/tools/pmc_summary.py $(dirname $(find $D -name "*counter_collection.csv" | head -1))
done 2>&1 | tee $OUT/summary.txt
