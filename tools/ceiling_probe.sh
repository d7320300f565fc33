#!/bin/bash
# GPU box: the ceiling probe (tools/ceiling_probe.py) plain and under one PMC pass each -> gpurun_out/ceiling/
set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/ceiling; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in real uniform; do
  timeout -k 10 120 python3 $R/tools/ceiling_probe.py $w 256 > $OUT/$w.log 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_$w -o p -- python3 $R/tools/ceiling_probe.py $w 256 > $OUT/pmc_$w.log 2>&1 || exit 1
  echo "== $w"; cat $OUT/$w.log; python3 $R/tools/pmc_summary.py $(dirname $(find $OUT/pmc_$w -name "*counter_collection.csv" | head -1)) | tee $OUT/pmc_${w}_summary.txt; tail -1 $OUT/pmc_$w.log | tee -a $OUT/pmc_${w}_summary.txt
done
