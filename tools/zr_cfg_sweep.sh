#!/bin/bash
# dev: frame rate for launch configurations of the z|r convolution that runs BESIDE the Gauss-Newton builder
OUT=gpurun_out/zc2.log; : > $OUT
for c in "" "1,12,16,2,2,4,1,3,1" "1,12,8,2,2,4,1,3,1" "1,16,16,2,2,4,1,3,1" "2,8,16,2,2,4,1,3,1" "1,16,8,2,2,4,1,3,1" "1,12,16,4,2,4,1,3,1" "1,16,16,4,2,4,1,3,1" "1,8,8,2,2,4,1,3,1" "1,8,16,1,2,4,1,3,1" "1,16,16,1,2,4,1,3,1" "1,12,16,2,2,4,1,3,2" ""; do
  CODD_GATE1_CFG=$c timeout 100 python bench.py --steps 100 --no-cpu-baseline --fp32-steps 0 --two-video-steps 0 --no-pmc-traffic 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('zr cfg [$c]', d['value'], d['ms_per_step'])" >> $OUT 2>/dev/null || echo "zr cfg [$c] FAILED" >> $OUT
done
cat $OUT
