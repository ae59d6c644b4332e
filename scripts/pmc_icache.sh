#!/bin/bash
# Instruction-cache counters of the scene04 trace launches (scripts/pmc_target.py trace): one rocprofv3 --pmc pass
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_icache; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*\|SQ_INSTS_BRANCH\|SQ_INSTS_SMEM[A-Z_]*\|SQC_DCACHE_[A-Z_]*" | sort -u | tr '\n' ' ' > $OUT/avail.txt
echo "available: $(cat $OUT/avail.txt)"
timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/run -- python3 $GRAFT_REPO_ROOT/scripts/pmc_target.py trace > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
python3 - <<PY
import csv,glob,collections
f=glob.glob("$OUT/run/**/*counter_collection.csv",recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"][:40]
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in acc.items():
    if "k_trace" in k: print(k, dict(v))
PY
