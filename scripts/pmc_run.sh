#!/bin/bash
# usage: pmc_run.sh <outdir-under-gpurun_out> <pmc_target.py arguments...>      e.g. pmc_run.sh pmc_trace trace
# One rocprofv3 run per counter group (no trace domains mixed in), each under its own time limit.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
run() {  # name, counters...
  local name=$1; shift
  # (a line a minute while the pass runs: the pass itself prints only at its end)
  ( while sleep 50; do echo "  pass $name of $ARGS running"; done ) & local hb=$!
  timeout -s KILL 280 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/scripts/pmc_target.py $ARGS > $OUT/$name.log 2>&1
  local rc=$?
  kill $hb 2> /dev/null; wait $hb 2> /dev/null
  if [ $rc -ne 0 ]; then echo "pass $name failed"; tail -3 $OUT/$name.log; return 1; fi
  echo "pass $name ok"
}
ARGS="$*"
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU &&
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT &&
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE &&
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum &&
run fetch FETCH_SIZE &&
run write WRITE_SIZE
