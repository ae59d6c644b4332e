#!/bin/bash
# gpu_regress.sh <lib A> <lib B>: two library builds under ab/ (e.g. r02 = the previous round's, main = this round's), twice each
# and interleaved, over the workloads that are not the headline: soups of 10 K .. 10 M triangles (trees in LDS, as plain
# records, as child lines), scene04 with its tree read from global memory, and scene01-03 at 1162x1008 (128-frame launches)
for lib in $1 $2 $1 $2; do
  echo "== $lib"
  PTAMD_LIB=ab/$lib.so FRAMES=16 timeout -k 10 120 python scripts/gpu_soup_sizes.py 10000 100000 300000 2>&1 | grep "^n " | awk '{print $1,$2,$(NF-7),$(NF-6),$(NF-5),$(NF-4),$(NF-3)}' || exit 1
  PTAMD_LIB=ab/$lib.so timeout -k 10 120 python scripts/gpu_soup_sizes.py 1000000 10000000 2>&1 | grep "^n " | awk '{print $1,$2,$(NF-7),$(NF-6),$(NF-5),$(NF-4),$(NF-3)}' || exit 1
  PTAMD_LIB=ab/$lib.so PTAMD_NO_LDS_TREE=1 timeout -k 10 100 python scripts/gpu_order.py scene04 1162 1008 128 | tail -1 | awk '{print "scene04 no-lds-tree", $(NF-1), $NF}' || exit 1
  for sc in scene01 scene02 scene03; do PTAMD_LIB=ab/$lib.so timeout -k 10 100 python scripts/gpu_order.py $sc 1162 1008 128 | tail -1 | awk '{print $1, $(NF-1), $NF}' || exit 1; done
done
