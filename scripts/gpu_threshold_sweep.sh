#!/bin/bash
# Wave-vote thresholds of k_trace_dual, one at a time around the defaults (scene04, 128-frame launches, scripts/gpu_order.py):
# lanes at a leaf that start a triangle-test round, parked slots that start a refill round, lanes without a ray that do,
# lanes with a ready parked ray that start a swap
run() { echo -n "$1  "; env $1 timeout -k 10 60 python scripts/gpu_order.py scene04 1162 1008 128 | tail -1 | cut -d' ' -f7- || exit 1; }
run PTAMD_NONE=0
for v in 4 6 12 16; do run PTAMD_DUAL_LEAF_T=$v; done
for v in 32 40 56 60; do run PTAMD_SHADE_T=$v; done
for v in 4 6 12 16; do run PTAMD_DUAL_REFILL_T=$v; done
for v in 2 3 6 8; do run PTAMD_SWAP_T=$v; done
run PTAMD_NONE=0
