#!/bin/bash
# dev aid: autotune N times, keep every tune db with its frames/s under gpurun_out/tune_search/
mkdir -p gpurun_out/tune_search
for i in $(seq 1 ${1:-6}); do
  rm -f gpurun_out/tune_search/db_$i.json
  python bench.py --no-cpu-baseline --retune --tune-db gpurun_out/tune_search/db_$i.json 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print($i, d['value'], d['roofline']['frac'])" | tee -a gpurun_out/tune_search/results.txt
done
