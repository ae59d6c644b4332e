#!/bin/bash
# Runs on the GPU box (gpurun -- 'bash scripts/gpu_collect_profiles.sh'): every r02_* measurement record kept under
# profiles/, written to gpurun_out/profiles_r02/ (copy from there into profiles/ and commit).
#   r02_bench_default.json            the JSON line of `python bench.py` (run under the profiler: next file)
#   r02_bench_default_kernel_stats.csv   rocprofv3 --kernel-trace --stats of that same run
#   r02_trace_pmc.json / r02_build_pmc.json / r02_soup10m_pmc.json   hardware counters, one rocprofv3 --pmc run per group
#   r02_bench_scene0{2,3}_4k.json     3840x2160, 512 spp per GPU (one GPU's share of the 4096 spp / 8 GPU configuration)
#   r02_soup10m_morton_pmc.json       the same soup with the tree over the Morton codes, nearest child first
#   r02_bench_soup100m*.json          the largest configuration: the reference's tree in both traversal orders, and
#                                     (_morton) the mode's defaults: tree over the Morton codes, nearest child first
#   r02_random_records.txt            random record read rate of the memory system (scripts/micro/random_records.hip)
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profiles_r02; mkdir -p $O; export TMPDIR=/tmp
cd $R
# counters first: bench.py reads profiles/r02_*_pmc.json of the same sources
bash scripts/pmc_run.sh pmc_trace trace && python3 scripts/pmc_summarize.py gpurun_out/pmc_trace $O/r02_trace_pmc.json "bash scripts/pmc_run.sh pmc_trace trace" > /dev/null || exit 1
bash scripts/pmc_run.sh pmc_build build && python3 scripts/pmc_summarize.py gpurun_out/pmc_build $O/r02_build_pmc.json "bash scripts/pmc_run.sh pmc_build build" > /dev/null || exit 1
bash scripts/pmc_run.sh pmc_soup soup 10000000 && python3 scripts/pmc_summarize.py gpurun_out/pmc_soup $O/r02_soup10m_pmc.json "bash scripts/pmc_run.sh pmc_soup soup 10000000" > /dev/null || exit 1
bash scripts/pmc_run.sh pmc_soup_morton soup 10000000 near morton && python3 scripts/pmc_summarize.py gpurun_out/pmc_soup_morton $O/r02_soup10m_morton_pmc.json "bash scripts/pmc_run.sh pmc_soup_morton soup 10000000 near morton" > /dev/null || exit 1
cp $O/r02_trace_pmc.json $O/r02_build_pmc.json profiles/
cd /tmp && rm -rf /tmp/kstats
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstats -- python3 $R/bench.py > $O/r02_bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 1; }
cp $(find /tmp/kstats -name '*kernel_stats.csv' | head -1) $O/r02_bench_default_kernel_stats.csv
cd $R
for sc in scene02_4k scene03_4k; do
  timeout -k 10 300 python3 bench.py --config $sc --steps 4 --warmup 1 --no-build --no-cpu-baseline > $O/r02_bench_$sc.json 2> $O/bench_4k.err || { tail -5 $O/bench_4k.err; exit 1; }
done
timeout -k 10 300 python3 bench.py --config soup100m --steps 3 --warmup 1 --topology index --near-first 0 > $O/r02_bench_soup100m.json 2> $O/bench_soup.err || { tail -5 $O/bench_soup.err; exit 1; }
timeout -k 10 300 python3 bench.py --config soup100m --steps 3 --warmup 1 --topology index --near-first 1 > $O/r02_bench_soup100m_near_first.json 2>> $O/bench_soup.err || { tail -5 $O/bench_soup.err; exit 1; }
timeout -k 10 300 python3 bench.py --config soup100m --steps 3 --warmup 1 > $O/r02_bench_soup100m_morton.json 2>> $O/bench_soup.err || { tail -5 $O/bench_soup.err; exit 1; }
( cd scripts/micro && for cfg in "160 64 3 1024" "1280 64 3 1024" "1280 128 6 1024" "6400 64 3 1024" "12800 64 3 1024"; do timeout -k 5 60 ./random_records $cfg 1000 || exit 1; done ) > $O/r02_random_records.txt
echo collected; ls -la $O
