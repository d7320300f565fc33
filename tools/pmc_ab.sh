#!/bin/bash
# PMC A/B of path-kernel configurations on the GPU box: tools/pmc_ab.sh 04 2r 2q ...  -> gpurun_out/pmc_ab/<cfg>/
R=$GRAFT_REPO_ROOT; which=$1; shift
cd /tmp && export TMPDIR=/tmp
for cfgx in "$@"; do
  cfg=${cfgx%%/*}; extra=${cfgx#*/}; [ "$extra" = "$cfgx" ] && extra=""      # cfg[/spp skip defs threshold]
  OUT=$R/gpurun_out/pmc_ab/${which}_$(echo $cfgx | tr '/ ,' '___'); rm -rf $OUT; mkdir -p $OUT
  timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT -o p -- python3 $R/tools/perf_cfg.py $which $cfg $extra > $OUT/run.log 2>&1 || exit 1
  echo "== $which $cfgx"; tail -1 $OUT/run.log; python3 $R/tools/pmc_summary.py $(dirname $(find $OUT -name "*counter_collection.csv" | head -1))
done
