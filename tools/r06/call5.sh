#!/bin/bash
# GPU box, round 6 call 5: the conditioned parity tests on the full goldens, the launch configurations the shipped db lacks
# (tools/extend_tune_db.py -> gpurun_out/mi355x_extended.json), the XCD barrier microbenchmark (double-buffered records), two
# default bench runs on an otherwise idle host.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_call5; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_headline_parity.py -q -s -k conditioned 2>&1 | grep "frame\|per-frame\|passed\|failed\|meet\|oracle-vs\|Error" | cut -c1-300 > $O/conditioned_parity.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/xcd_barrier tools/ubench/xcd_barrier.hip > $O/xcd_barrier.log 2>&1 && timeout 120 /tmp/xcd_barrier 2000 >> $O/xcd_barrier.log 2>&1
B="python bench.py --no-cpu-baseline --no-pmc-traffic --two-video-steps 0 --fp32-steps 0 --steps 100"
{ for i in 1 2; do echo "== default"; $B 2>/dev/null | cut -c1-1200; done; } > $O/bench.log
timeout 900 python tools/extend_tune_db.py > $O/extend_tune_db.log 2>&1
cp $R/gpurun_out/mi355x_extended.json $O/ 2>/dev/null
grep "per-frame mean\|meet\|passed\|failed" $O/conditioned_parity.log | cut -c1-400; cat $O/xcd_barrier.log; cut -c1-200 $O/bench.log; tail -3 $O/extend_tune_db.log
