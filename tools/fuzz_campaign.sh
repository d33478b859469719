#!/bin/bash
# the parity fuzz campaigns of a round on the GPU box, one JSON line each into gpurun_out/<tag>_fuzz.jsonl
#   tools/fuzz_campaign.sh <tag> [scale]      scale 1 = the sizes below, 0.1 = a tenth of them
set -o pipefail
tag=${1:-fuzz}; k=${2:-1}
out=gpurun_out/${tag}_fuzz.jsonl
: > "$out"
n() { python3 -c "print(max(20, int($1 * $k)))"; }
run() { echo "== $*" >&2; timeout -k 10 900 python3 -m tests.fuzz_parity "$@" 2>/dev/null | tail -1 >> "$out" || { echo "FAILED: $*" >&2; exit 1; }; tail -1 "$out" >&2; }
run --cases $(n 20000) --seed 31
run --cases $(n 6000) --seed 32 --wide
run --data-aware $(n 2000) --seed 33
run --gptq $(n 1500) --seed 34
run --hessian $(n 300) --seed 35
run --reducers $(n 2000) --seed 36
run --errors $(n 2000) --seed 37
run --decompress $(n 3000) --seed 38
run --misc $(n 2000) --seed 39
echo done >&2
