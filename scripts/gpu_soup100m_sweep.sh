#!/bin/bash
# gpu_soup100m_sweep.sh VAR=value ...: the 100M-triangle soup (bench.py --config soup100m, parity mode and the mode's defaults)
# once per environment setting; "X=0" is the baseline
for e in "$@"; do
  for a in "--topology index --near-first 0" ""; do
    env $e timeout -k 10 300 python bench.py --config soup100m --steps 3 --warmup 1 $a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$e', '${a:-defaults}', round(d['value'],2), 'Mrays/s', round(d['ms_per_step'],1), 'ms')" || exit 1
  done
done
