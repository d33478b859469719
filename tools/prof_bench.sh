#!/bin/bash
# Final profiles of the default bench (run on the GPU box from the repository root): the plain line, the same command
# under rocprofv3 --kernel-trace --stats, and the two --pmc passes for the HBM traffic.
#   tools/prof_bench.sh <round tag, e.g. r3>
set -e
tag=${1:-r3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R && timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/${tag}_bench.json 2> $O/${tag}_bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_kt -o kt -- python3 $R/bench.py --no-cpu-baseline > $O/${tag}_bench_under_rocprof.json 2> $O/${tag}_kt.err
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${tag}_fetch -o f -- python3 $R/bench.py --steps 2 --warmup 1 --sustained-s 0 --no-cpu-baseline > $O/${tag}_fetch.out 2> $O/${tag}_fetch.err
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${tag}_write -o w -- python3 $R/bench.py --steps 2 --warmup 1 --sustained-s 0 --no-cpu-baseline > $O/${tag}_write.out 2> $O/${tag}_write.err
cd $R
cp $(find $O/${tag}_kt -name "*kernel_stats.csv" | head -1) $O/${tag}_bench_kernel_stats.csv
cp $(find $O/${tag}_fetch -name "*counter_collection.csv" | head -1) $O/${tag}_pmc_FETCH_SIZE.csv
cp $(find $O/${tag}_write -name "*counter_collection.csv" | head -1) $O/${tag}_pmc_WRITE_SIZE.csv
rm -rf $O/${tag}_kt $O/${tag}_fetch $O/${tag}_write
head -3 $O/${tag}_bench_kernel_stats.csv | cut -c1-200
cat $O/${tag}_bench.json | cut -c1-1500
