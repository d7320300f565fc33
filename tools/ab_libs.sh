#!/bin/bash
# same-box A/B of two builds of the library (tools/ab/librt_old.so vs the tree's), interleaved: old new old new
for rep in 1 2; do
  for which in old new; do
    if [ "$which" = old ]; then export RT_MI355_LIB=$PWD/tools/ab/librt_old.so; else unset RT_MI355_LIB; fi
    echo "== $which (round $rep)"
    timeout -k 10 120 python3 -u tools/time_cfg.py "$@" || exit 1
  done
done
