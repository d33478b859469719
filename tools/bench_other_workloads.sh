#!/bin/bash
# The non-headline RTN workloads of bench.py at full layer count, one JSON object keyed by workload (GPU box).
#   tools/bench_other_workloads.sh <out.json>
out=${1:-gpurun_out/bench_other_workloads.json}
echo "{" > $out
first=1
for w in llama3-8b-int4asym-g128 llama3-8b-int8asym llama3-70b-int4sym-g128 mixtral-nf4-g32 mixtral-mxfp4-g32; do
  line=$(timeout -k 10 200 python3 bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1) || exit 1
  [ $first = 1 ] || echo "," >> $out
  first=0
  echo "\"$w\": $line" >> $out
done
echo "}" >> $out
python3 -c "import json,sys; d=json.load(open('$out')); [print(k, d[k]['value'], 'GB/s', d[k]['roofline']['frac'], 'sustained', d[k]['sustained']['roofline_frac']) for k in d]"
