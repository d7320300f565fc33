#!/bin/bash
# unit-size sweep at steady clocks (tools/time_cfg.py): one process per setting
for k in default 2 3 4 5 6; do
  if [ "$k" = default ]; then unset RT_MI355_UNIT_LOG2; else export RT_MI355_UNIT_LOG2=$k; fi
  echo "== unit_log2 $k"
  timeout -k 10 120 python3 -u tools/time_cfg.py 03 04q 04 || exit 1
done
