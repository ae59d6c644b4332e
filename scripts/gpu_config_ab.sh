#!/bin/bash
# gpu_config_ab.sh <lib A> <lib B> "<scene W H frames>" ...: launch times of two library builds under ab/ per configuration
# (environment variables apply to both)
a=$1; b=$2; shift 2
for c in "$@"; do
  for lib in $a $b; do
    echo -n "$lib  "
    env PTAMD_LIB=ab/$lib.so timeout -k 10 200 python -u scripts/gpu_order.py $c | tail -1 | cut -c1-150 || exit 1
  done
done
