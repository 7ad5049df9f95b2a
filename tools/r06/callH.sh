#!/bin/bash
# GPU box (no GPU work): hostB evaluations of the conditioned oracle on the box's host CPU, ONE process at a time (concurrent
# oracle processes starve each other on this host: 259 s instead of 22 s per frame in calls 3 / 4 / A).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_callH; mkdir -p $O; cd $R
G=tests/golden/make_conditioned_golden.py; export CODD_GOLDEN_STATE_DIR=/tmp CODD_GOLDEN_THREADS=32
CODD_GOLDEN_OUT=$O/box_main5.npz timeout 420 python $G cfg5_16c > $O/box_main_cfg5.log 2>&1
CODD_GOLDEN_OUT=$O/box_alt5.npz CODD_GOLDEN_VARIANT=nomkldnn timeout 480 python $G cfg5_16c > $O/box_alt_cfg5.log 2>&1
CODD_GOLDEN_OUT=$O/box_main3.npz CODD_GOLDEN_FRAMES=${N3:-8} timeout ${T3:-540} python $G cfg3_50c > $O/box_main_cfg3.log 2>&1
rm -f $O/*.tmp*; tail -1 $O/box_*.log | cut -c1-150
