#!/bin/bash
# gpu_env_sweep.sh <library under ab/> <VARIABLE> <value> ...: scene04 128-frame launch times (scripts/gpu_order.py) with
# the environment variable set to each value in turn
lib=$1; var=$2; shift 2
for v in "$@"; do
  echo -n "$lib $var=$v  "
  env PTAMD_LIB=ab/$lib.so $var=$v timeout -k 10 120 python -u scripts/gpu_order.py scene04 1162 1008 ${FRAMES:-128} | tail -1 || exit 1
done
