#!/bin/bash
# Runs on the GPU box (via gpurun): the bench line, a kernel-trace pass and four PMC passes (each its own run, as
# the MI355X guide prescribes), all under gpurun_out/prof/.  tools/make_profile_json.py turns them into profiles/.
set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o kt -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_under_trace.json 2> $OUT/trace.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_sq -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_sq.json 2> $OUT/pmc_sq.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_write -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_write.json 2> $OUT/pmc_write.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT --output-format csv -d $OUT/pmc_flops -o p -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_flops.json 2> $OUT/pmc_flops.err || exit 1
# 03 (config 3): lane utilisation and instruction counts of k_path_jit<3> (tools/perf_cfg.py 03)
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_03 -o p -- python3 $R/tools/perf_cfg.py 03 1r 256 > $OUT/pmc_03.log 2>&1 || exit 1
python3 $R/tools/pmc_summary.py $(dirname $(find $OUT/pmc_03 -name "*counter_collection.csv" | head -1)) > $OUT/pmc_03_summary.txt; cat $OUT/pmc_03.log | tail -2 >> $OUT/pmc_03_summary.txt
find $OUT -name "*.csv" | head -20
