#!/bin/bash
# same-box A/B of an environment switch, interleaved: tools/ab_env.sh VAR=value case...   (default, switch, default, switch)
kv=$1; shift
for rep in 1 2; do
  for which in default "$kv"; do
    echo "== $which (round $rep)"
    if [ "$which" = default ]; then timeout -k 10 120 python3 -u tools/time_cfg.py "$@" || exit 1
    else env "$kv" timeout -k 10 120 python3 -u tools/time_cfg.py "$@" || exit 1; fi
  done
done
