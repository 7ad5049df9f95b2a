#!/bin/bash
# (historical: ran on the tree BEFORE the prune -- the CODD_HR_FUSE_* / CODD_CONVB_RING2 / CODD_GN_Q4 switches it sets were deleted afterwards)
# GPU box, round 6 call 1: does the product meet rule (1) on the conditioned cfg5_16c golden -- by default and with the
# re-ordering switches that the ill-conditioned goldens vetoed?  + same-lease A/B of their frame rate.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_call1; mkdir -p $O; cd $R
F="frame\|worst\|per-frame\|passed\|failed\|golden\|held\|meet\|Error\|assert"
T=tests/test_gpu_headline_parity.py
python -m pytest $T -q -s -k "conditioned" 2>&1 | grep "$F" | cut -c1-300 > $O/cond_default.log
CODD_GN_Q4=256 python -m pytest $T -q -s -k "conditioned" 2>&1 | grep "$F" | cut -c1-300 > $O/cond_CODD_GN_Q4_256.log
for SW in CODD_HR_FUSE_SUM CODD_HR_FUSE_TERMS; do
  env $SW=1 python -m pytest $T -q -s -k "conditioned or recurrent" 2>&1 | grep "$F" | cut -c1-300 > $O/cond_$SW.log
done
B="python bench.py --no-cpu-baseline --no-pmc-traffic --two-video-steps 0 --fp32-steps 0 --steps 100"
{ for i in 1 2; do
    echo "== default";            $B 2>/dev/null | cut -c1-200
    echo "== CODD_HR_FUSE_SUM=1";   CODD_HR_FUSE_SUM=1 $B 2>/dev/null | cut -c1-200
    echo "== CODD_HR_FUSE_TERMS=1"; CODD_HR_FUSE_TERMS=1 $B 2>/dev/null | cut -c1-200
    echo "== CODD_CONVB_RING2=1";   CODD_CONVB_RING2=1 $B 2>/dev/null | cut -c1-200
  done; } > $O/ab_hr_fuse.log
nproc > $O/host.log; lscpu | grep "Model name\|^CPU(s)\|Thread\|Socket" >> $O/host.log
tail -3 $O/cond_*.log; cat $O/ab_hr_fuse.log
