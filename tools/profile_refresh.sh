#!/bin/bash
# GPU-box job: rocprofv3 kernel stats + HBM counters (separate --pmc passes) of the default bench, bench JSON itself
set -e
O=gpurun_out/prof_refresh
mkdir -p $O
export TMPDIR=/tmp
python bench.py > $O/bench_default.json 2> $O/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 bench.py --steps 1280 --warmup 256 --no-cpu-baseline > $O/bench_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/pmc_fetch -o run -- python3 bench.py --steps 256 --warmup 64 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/pmc_write -o run -- python3 bench.py --steps 256 --warmup 64 --no-cpu-baseline > /dev/null 2>&1
python tools/pmc_summary.py $O/pmc_fetch k_rollout2 > $O/pmc_fetch.txt
python tools/pmc_summary.py $O/pmc_write k_rollout2 > $O/pmc_write.txt
cat $O/pmc_fetch.txt $O/pmc_write.txt
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs head -4
cat $O/bench_default.json
