#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and MFMA utilisation per kernel for a short
# serial-stream bench run.  Run on the GPU box: tools/pmc_hbm.sh ; writes gpurun_out/pmc_r1/*.csv
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_r1; mkdir -p $OUT
CMD="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --serial-streams --steps 4 --prewarm 2 --warmup 1 --no-graph"
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/pm_$tag
  timeout 280 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pm_$tag -o p -- $CMD > /dev/null 2>&1
  cp /tmp/pm_$tag/p_counter_collection.csv $OUT/${tag}_counter_collection.csv
  cp /tmp/pm_$tag/p_kernel_trace.csv $OUT/${tag}_kernel_trace.csv
done
ls -la $OUT
