#!/bin/bash
# dev (GPU box): kernel timeline of one stereo-only frame (graph replay) -> gpurun_out/stereo_timeline/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/stereo_timeline; mkdir -p $OUT
rm -rf /tmp/st_so
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_so -o s -- python $R/bench.py --stereo-only --steps 40 --prewarm 20 --no-cpu-baseline --fp32-steps 0 --two-video-steps 0 --no-pmc-traffic ${EXTRA:-} > $OUT/bench.log 2>&1
cp /tmp/st_so/s_kernel_stats.csv $OUT/kernel_stats.csv
python3 $R/tools/timeline_gaps.py /tmp/st_so/s_kernel_trace.csv --dump $OUT/frame_sequence.txt > $OUT/timeline.txt 2>&1
tail -1 $OUT/bench.log | cut -c1-300; head -30 $OUT/timeline.txt
