#!/bin/bash
# rocprofv3 kernel-trace + stats of a python target on the GPU box, summary into gpurun_out/<name>_kernel_stats.csv
#   tools/prof.sh <name> <script.py> [args...]       (run from the repository root on the box)
set -o pipefail
name=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$name
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o "$name" -- python3 "$root/$1" "${@:2}" > "$out/stdout.txt" 2> "$out/stderr.txt"
rc=$?
cd "$root"
f=$(find "$out" -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" "gpurun_out/${name}_kernel_stats.csv"; head -${PROF_LINES:-16} "$f" | cut -c1-230; fi
rm -f "$out"/*/*kernel_trace.csv "$out"/*kernel_trace.csv 2>/dev/null
tail -3 "$out/stdout.txt" | cut -c1-1500
exit $rc
