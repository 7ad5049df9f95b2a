#!/bin/bash
# GPU box: the conditioned parity tests (whatever the committed golden holds) + hostB evaluations of the conditioned oracle for
# the call's duration (HOSTB = "case frames threads" triples; resume files in /tmp)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_${TAG:-callA}; mkdir -p $O; cd $R
G=tests/golden/make_conditioned_golden.py; export CODD_GOLDEN_STATE_DIR=/tmp
set -- ${HOSTB:-cfg5_16c 16 24}
while [ $# -ge 3 ]; do
  ( CODD_GOLDEN_OUT=$O/box_main_$1.npz CODD_GOLDEN_THREADS=$3 CODD_GOLDEN_FRAMES=$2 python $G $1 > $O/box_main_$1.log 2>&1 ) &
  ( CODD_GOLDEN_OUT=$O/box_alt_$1.npz CODD_GOLDEN_VARIANT=nomkldnn CODD_GOLDEN_THREADS=$3 CODD_GOLDEN_FRAMES=$2 python $G $1 > $O/box_alt_$1.log 2>&1 ) &
  shift 3
done
sleep 3
python -m pytest tests/test_gpu_headline_parity.py -q -s -k "${KEXPR:-conditioned}" 2>&1 | grep "frame\|per-frame\|passed\|failed\|meet\|oracle-vs\|Error" | cut -c1-300 > $O/conditioned_parity.log
grep "per-frame mean\|meet\|passed\|failed" $O/conditioned_parity.log | cut -c1-600
wait
rm -f $O/*.tmp*; tail -1 $O/box_*.log | cut -c1-150
