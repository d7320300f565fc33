#!/bin/bash
# What do the SQ counters read for a stream that is KNOWN to keep the VALU 100 % busy?  tools/clock_probe: every wave a chain of independent v_fma, 8 waves per SIMD
# (sustains 93 % of the FP32 peak).  The same counters on the path kernel (tools/perf_one.py 04) then give its VALU utilisation by ratio.
set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/calib; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/fma -o p -- $R/tools/clock_probe > $OUT/fma.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/k04 -o p -- python3 $R/tools/perf_one.py 04 > $OUT/k04.log 2>&1 || exit 1
for w in fma k04; do echo "== $w"; PMC_KERNEL=$([ $w = fma ] && echo k_load || echo k_path) python3 $R/tools/pmc_summary.py $(dirname $(find $OUT/$w -name "*counter_collection.csv" | head -1)); tail -2 $OUT/$w.log; done
