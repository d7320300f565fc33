#!/bin/bash
# Sample slices per frame (RT_MI355_SLICES) vs kernel time: shorter blocks shorten the tail of the launch.
cd $GRAFT_REPO_ROOT
for cfg in "03 1r 256" "04 1r 256"; do
  for s in 0 2 4 8 16 32; do
    if [ $s = 0 ]; then unset RT_MI355_SLICES; else export RT_MI355_SLICES=$s; fi
    echo "slices ${s} (0 = library default): $(timeout -k 10 120 python3 tools/perf_cfg.py $cfg 2>/dev/null | tail -1)"
  done
done
