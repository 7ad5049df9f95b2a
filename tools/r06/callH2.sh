#!/bin/bash
# GPU box (no GPU work): more frames of the hostB evaluations of cfg3_50c, one process at a time (see callH.sh)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_callH2; mkdir -p $O; cd $R
G=tests/golden/make_conditioned_golden.py; export CODD_GOLDEN_STATE_DIR=/tmp CODD_GOLDEN_THREADS=32
CODD_GOLDEN_OUT=$O/box_${V:-main}3.npz CODD_GOLDEN_VARIANT=${VARIANT:-} CODD_GOLDEN_FRAMES=${N3:-19} timeout ${T3:-1740} python $G cfg3_50c > $O/box_${V:-main}_cfg3.log 2>&1
rm -f $O/*.tmp*; tail -1 $O/box_*.log | cut -c1-150
