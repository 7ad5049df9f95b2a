#!/bin/bash
# GPU box, round 6 call 3: the pruned / re-plumbed library on the GPU -- full GPU test suite, the tuner's per-candidate
# table for the update-block layers (incl. the new 96-channel / 20-unit instantiations), the XCD barrier microbenchmark,
# two default bench runs -- while the box's HOST CPU evaluates the conditioned oracle (hostB variants; resume files in /tmp:
# gpurun_out/ merges back at most 64 MiB).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_call3; mkdir -p $O; cd $R
G=tests/golden/make_conditioned_golden.py; export CODD_GOLDEN_STATE_DIR=/tmp
( CODD_GOLDEN_OUT=$O/box_main5.npz CODD_GOLDEN_THREADS=24 python $G cfg5_16c > $O/box_main_cfg5.log 2>&1 ) &
( CODD_GOLDEN_OUT=$O/box_alt5.npz CODD_GOLDEN_VARIANT=nomkldnn CODD_GOLDEN_THREADS=24 python $G cfg5_16c > $O/box_alt_cfg5.log 2>&1 ) &
( CODD_GOLDEN_OUT=$O/box_main3.npz CODD_GOLDEN_THREADS=40 CODD_GOLDEN_FRAMES=${HOSTB_FRAMES:-14} python $G cfg3_50c > $O/box_main_cfg3.log 2>&1 ) &
( CODD_GOLDEN_OUT=$O/box_alt3.npz CODD_GOLDEN_VARIANT=nomkldnn CODD_GOLDEN_THREADS=40 CODD_GOLDEN_FRAMES=${HOSTB_FRAMES:-14} python $G cfg3_50c > $O/box_alt_cfg3.log 2>&1 ) &
sleep 15
python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/gpu_tests.log
python -m pytest tests/test_gpu_headline_parity.py -q -s -k conditioned 2>&1 | grep "frame\|per-frame\|passed\|failed\|meet\|oracle-vs\|Error" | cut -c1-260 > $O/conditioned_parity.log
timeout 900 python tools/sweep_update_block.py > $O/sweep_update_block.log 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/xcd_barrier tools/ubench/xcd_barrier.hip > $O/xcd_barrier.log 2>&1 && timeout 120 /tmp/xcd_barrier 2000 >> $O/xcd_barrier.log 2>&1
B="python bench.py --no-cpu-baseline --no-pmc-traffic --two-video-steps 0 --fp32-steps 0 --steps 100"
{ for i in 1 2; do echo "== default"; $B 2>/dev/null | cut -c1-200; done; } > $O/bench.log
wait
rm -f $O/*.tmp*
tail -6 $O/gpu_tests.log; cat $O/xcd_barrier.log; cat $O/bench.log; tail -2 $O/box_*.log; du -sh $R/gpurun_out
