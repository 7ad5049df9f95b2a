#!/bin/bash
# GPU box, round 6 call 3: the pruned / re-plumbed library on the GPU -- full GPU test suite, the tuner's per-candidate
# table for the update-block layers (incl. the new 96-channel / 20-unit instantiations), the XCD barrier microbenchmark,
# and two default bench runs.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_call3; mkdir -p $O; cd $R
python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/gpu_tests.log
timeout 900 python tools/sweep_update_block.py > $O/sweep_update_block.log 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/xcd_barrier tools/ubench/xcd_barrier.hip > $O/xcd_barrier.log 2>&1 && timeout 120 /tmp/xcd_barrier 2000 >> $O/xcd_barrier.log 2>&1
B="python bench.py --no-cpu-baseline --no-pmc-traffic --two-video-steps 0 --fp32-steps 0 --steps 100"
{ for i in 1 2; do echo "== default"; $B 2>/dev/null | cut -c1-200; done; } > $O/bench.log
tail -5 $O/gpu_tests.log; cat $O/xcd_barrier.log; cat $O/bench.log; grep -c "us " $O/sweep_update_block.log
