#!/bin/bash
# gpu_item_sweep.sh "<coarse_items> <fine_items> <fine_chunk>" ...: scene04 launch times (scripts/gpu_order.py; FRAMES, SIZE
# from the environment, default 128 and "1162 1008") for k_trace_dual's item knobs
for c in "$@"; do
  set -- $c
  echo -n "long items per lane $1, short items per lane $2 of $3 frames (0: auto):  "
  env PTAMD_COARSE_ITEMS=$1 PTAMD_FINE_ITEMS=$2 PTAMD_FINE_CHUNK=$3 timeout -k 10 120 python -u scripts/gpu_order.py scene04 ${SIZE:-1162 1008} ${FRAMES:-128} | tail -1 | cut -d' ' -f7- || exit 1
done
