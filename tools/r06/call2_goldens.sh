#!/bin/bash
# GPU box, round 6 call 2: the conditioned goldens' oracle evaluations on the BOX's host CPU (2 x EPYC 9575F) -- the default
# (oneDNN) and the ATen im2col + sgemm evaluation of both cases, four processes side by side -- while the GPU runs the test suite.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_gold; mkdir -p $O; cd $R
G=tests/golden/make_conditioned_golden.py
( CODD_GOLDEN_OUT=$O/box_main.npz CODD_GOLDEN_THREADS=40 python $G cfg3_50c > $O/box_main_cfg3.log 2>&1 ) &
( CODD_GOLDEN_OUT=$O/box_alt.npz CODD_GOLDEN_VARIANT=nomkldnn CODD_GOLDEN_THREADS=40 python $G cfg3_50c > $O/box_alt_cfg3.log 2>&1 ) &
( CODD_GOLDEN_OUT=$O/box_main5.npz CODD_GOLDEN_THREADS=16 python $G cfg5_16c > $O/box_main_cfg5.log 2>&1 ) &
( CODD_GOLDEN_OUT=$O/box_alt5.npz CODD_GOLDEN_VARIANT=nomkldnn CODD_GOLDEN_THREADS=16 python $G cfg5_16c > $O/box_alt_cfg5.log 2>&1 ) &
sleep 20
python -m pytest tests -m gpu -q -x --deselect "tests/test_gpu_headline_parity.py::test_conditioned_sequence_meets_north_star_bound[cfg3_50c]" 2>&1 | tail -15 > $O/gpu_tests.log
wait
rm -f $O/*.state.pt $O/*.tmp*
tail -2 $O/*.log; ls -la $O
