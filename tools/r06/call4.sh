#!/bin/bash
# GPU box, round 6 call 4: the tuner's per-candidate table for the update-block layers, the XCD barrier microbenchmark and two
# default bench runs on the GPU (little host work) -- beside the hostB evaluations of the conditioned oracle on the box's CPU
# (nothing else CPU-heavy in this call: call 3's oracle runs were starved by the test suite's own host-side oracle recomputes).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_call4; mkdir -p $O; cd $R
G=tests/golden/make_conditioned_golden.py; export CODD_GOLDEN_STATE_DIR=/tmp
( CODD_GOLDEN_OUT=$O/box_main3.npz CODD_GOLDEN_THREADS=32 CODD_GOLDEN_FRAMES=${HOSTB_FRAMES:-12} timeout 1000 python $G cfg3_50c > $O/box_main_cfg3.log 2>&1 ) &
( CODD_GOLDEN_OUT=$O/box_alt3.npz CODD_GOLDEN_VARIANT=nomkldnn CODD_GOLDEN_THREADS=32 CODD_GOLDEN_FRAMES=${HOSTB_FRAMES:-12} timeout 1000 python $G cfg3_50c > $O/box_alt_cfg3.log 2>&1 ) &
( CODD_GOLDEN_OUT=$O/box_main5.npz CODD_GOLDEN_THREADS=16 timeout 1000 python $G cfg5_16c > $O/box_main_cfg5.log 2>&1 ) &
( CODD_GOLDEN_OUT=$O/box_alt5.npz CODD_GOLDEN_VARIANT=nomkldnn CODD_GOLDEN_THREADS=16 timeout 1000 python $G cfg5_16c > $O/box_alt_cfg5.log 2>&1 ) &
sleep 5
timeout 600 python tools/sweep_update_block.py > $O/sweep_update_block.log 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/xcd_barrier tools/ubench/xcd_barrier.hip > $O/xcd_barrier.log 2>&1 && timeout 120 /tmp/xcd_barrier 2000 >> $O/xcd_barrier.log 2>&1
B="python bench.py --no-cpu-baseline --no-pmc-traffic --two-video-steps 0 --fp32-steps 0 --steps 100"
{ for i in 1 2; do echo "== default"; $B 2>/dev/null | cut -c1-200; done; } > $O/bench.log
cat $O/xcd_barrier.log; cat $O/bench.log
wait
rm -f $O/*.tmp*
tail -2 $O/box_*.log; du -sh $R/gpurun_out
