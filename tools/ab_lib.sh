#!/bin/bash
# dev (GPU box): same-lease A/B of two builds of libcodd_hip.so: alternating bench.py runs, in-tree library vs $1
# usage: tools/ab_lib.sh ab/libcodd_hip_old.so [pairs] [steps]
old=$1; n=${2:-3}; steps=${3:-100}
for i in $(seq 1 $n); do
  a=$(CODD_LIB_AB=$PWD/$old python bench.py --steps $steps --no-cpu-baseline --no-pmc-traffic --fp32-steps 0 --two-video-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['ms_per_frame'], d['epe_vs_synthetic_gt'])")
  b=$(python bench.py --steps $steps --no-cpu-baseline --no-pmc-traffic --fp32-steps 0 --two-video-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['ms_per_frame'], d['epe_vs_synthetic_gt'])")
  echo "pair $i: old [$a]   new [$b]   (frames/s, conv_bf16 family ms/frame by HIP events, epe)"
done
