#!/bin/bash
# Compiler-option sweep for the scene-specialised kernel (RT_MI355_JIT_FLAGS), headline scene at 4K / 256 spp.
cd $GRAFT_REPO_ROOT
run() { export RT_MI355_JIT_FLAGS="$1"; echo "[$1] $(timeout -k 10 120 python3 tools/perf_cfg.py 04 1r 256 2>&1 | tail -1 | cut -c1-100)"; }
run ""
run "-mllvm;-amdgpu-sched-strategy=max-ilp"
run "-mllvm;-amdgpu-sched-strategy=max-memory-clause"
run "-mllvm;-amdgpu-sched-strategy=iterative-ilp"
run "-mllvm;-amdgpu-sched-strategy=iterative-minreg"
run "-O2"
run "-mllvm;-amdgpu-use-divergent-register-indexing"
run "-mllvm;-enable-post-misched=0"
run "-mllvm;-amdgpu-enable-max-ilp-scheduling-strategy"
run "-mllvm;-amdgpu-schedule-relaxed-occupancy"
run "-munsafe-fp-atomics"
run "-mllvm;-amdgpu-waitcnt-forcezero=0;-mllvm;-amdgpu-vgpr-index-mode"
run ""
