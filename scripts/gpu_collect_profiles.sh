#!/bin/bash
# Runs on the GPU box (gpurun -- 'bash scripts/gpu_collect_profiles.sh [part ...]'): every r03_* measurement record kept under
# profiles/, written to gpurun_out/profiles_r03/ (copy from there into profiles/ and commit).  Parts (default: all, in this order;
# a gpurun call is limited to 20 minutes, so the parts are usually spread over several calls):
#   pmc      r03_trace_pmc.json, r03_build_pmc.json, r03_soup10m_pmc.json, r03_soup10m_morton_pmc.json: hardware counters,
#            one rocprofv3 --pmc run per counter group (scripts/pmc_run.sh)
#   pmc4k    r03_trace_scene02_4k_pmc.json, r03_trace_scene03_4k_pmc.json: the same for scene02 / scene03 at 3840x2160
#   pmc100m  r03_soup100m_pmc.json, r03_soup100m_near_first_pmc.json, r03_soup100m_morton_pmc.json: the same for the
#            largest configuration (reference's tree in both traversal orders; tree over the codes, nearest child first);
#            pmc100m_index / pmc100m_near / pmc100m_morton: one of the three
#   bench    r03_bench_default.json + r03_bench_default_kernel_stats.csv (python bench.py under rocprofv3 --kernel-trace --stats),
#            r03_bench_scene0{2,3}_4k.json, r03_bench_soup100m*.json, r03_random_records.txt
# bench.py reads the counter summaries from profiles/: run the pmc parts first and copy their files there.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profiles_r03; mkdir -p $O; export TMPDIR=/tmp
cd $R
PARTS="${*:-pmc pmc4k pmc100m bench}"
pmc() {  # output name, pmc_target arguments...
  local name=$1; shift
  bash scripts/pmc_run.sh pmc_$name "$@" && python3 scripts/pmc_summarize.py gpurun_out/pmc_$name $O/r03_${name}_pmc.json "bash scripts/pmc_run.sh pmc_$name $*" > /dev/null || exit 1
  cp $O/r03_${name}_pmc.json profiles/
}
for part in $PARTS; do
case $part in
pmc)
  pmc trace trace
  pmc build build
  pmc soup10m soup 10000000
  pmc soup10m_morton soup 10000000 near morton
  ;;
pmc4k)
  pmc trace_scene02_4k trace scene02 3840 2160
  pmc trace_scene03_4k trace scene03 3840 2160
  ;;
pmc100m)
  pmc soup100m soup100m index
  pmc soup100m_near_first soup100m index near
  pmc soup100m_morton soup100m morton near
  ;;
pmc100m_index) pmc soup100m soup100m index ;;
pmc100m_near) pmc soup100m_near_first soup100m index near ;;
pmc100m_morton) pmc soup100m_morton soup100m morton near ;;
bench)
  cd /tmp && rm -rf /tmp/kstats
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstats -- python3 $R/bench.py > $O/r03_bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 1; }
  cp $(find /tmp/kstats -name '*kernel_stats.csv' | head -1) $O/r03_bench_default_kernel_stats.csv
  cd $R
  for sc in scene02_4k scene03_4k; do
    timeout -k 10 300 python3 bench.py --config $sc --steps 4 --warmup 1 --no-build --no-cpu-baseline > $O/r03_bench_$sc.json 2> $O/bench_4k.err || { tail -5 $O/bench_4k.err; exit 1; }
  done
  timeout -k 10 300 python3 bench.py --config soup100m --steps 3 --warmup 1 --topology index --near-first 0 > $O/r03_bench_soup100m.json 2> $O/bench_soup.err || { tail -5 $O/bench_soup.err; exit 1; }
  timeout -k 10 300 python3 bench.py --config soup100m --steps 3 --warmup 1 --topology index --near-first 1 > $O/r03_bench_soup100m_near_first.json 2>> $O/bench_soup.err || { tail -5 $O/bench_soup.err; exit 1; }
  timeout -k 10 300 python3 bench.py --config soup100m --steps 3 --warmup 1 > $O/r03_bench_soup100m_morton.json 2>> $O/bench_soup.err || { tail -5 $O/bench_soup.err; exit 1; }
  ( cd scripts/micro && for cfg in "160 64 3 1024" "1280 64 3 1024" "1280 128 6 1024" "6400 64 3 1024" "12800 64 3 1024" "12800 128 6 1024"; do timeout -k 5 60 ./random_records $cfg 1000 || exit 1; done ) > $O/r03_random_records.txt
  ;;
*) echo "unknown part $part"; exit 1 ;;
esac
done
echo collected; ls -la $O
