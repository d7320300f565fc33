#!/bin/bash
# Same-box A/B of library builds: tools/ab.sh A B C  (each tools/ab/<name>.so), two interleaved rounds.
for round in 1 2; do
  for v in "$@"; do
    echo "== $v (round $round)"
    RT_MI355_LIB=$PWD/tools/ab/$v.so timeout -k 10 120 python -u tools/perf.py 0 || exit 1
  done
done
