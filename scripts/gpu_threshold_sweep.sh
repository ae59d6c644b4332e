for lt in 6 10 12; do PTAMD_DUAL_LEAF_T=$lt timeout -k 10 60 python scripts/gpu_order.py scene04 1162 1008 128 | tail -1 | sed "s/^/leaf_t $lt /"; done
for st in 40 56; do PTAMD_SHADE_T=$st timeout -k 10 60 python scripts/gpu_order.py scene04 1162 1008 128 | tail -1 | sed "s/^/shade_t $st /"; done
for rt in 2 8; do PTAMD_DUAL_REFILL_T=$rt timeout -k 10 60 python scripts/gpu_order.py scene04 1162 1008 128 | tail -1 | sed "s/^/starve_t $rt /"; done
for sw in 2 8; do PTAMD_SWAP_T=$sw timeout -k 10 60 python scripts/gpu_order.py scene04 1162 1008 128 | tail -1 | sed "s/^/swap_t $sw /"; done
for ff in "8 8" "16 8" "12 4" "12 16"; do set -- $ff; PTAMD_FINE_ITEMS=$1 PTAMD_FINE_CHUNK=$2 timeout -k 10 60 python scripts/gpu_order.py scene04 1162 1008 128 | tail -1 | sed "s/^/fine $1 $2 /"; done
timeout -k 10 60 python scripts/gpu_order.py scene04 1162 1008 128 | tail -1 | sed "s/^/default /"
