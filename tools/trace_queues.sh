#!/bin/bash
# dev (GPU box): kernel trace of the frame graph with queue / stream ids -> gpurun_out/trace_queues/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/trace_queues; mkdir -p $OUT
rm -rf /tmp/tq
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tq -o s -- python $R/bench.py --steps 20 --prewarm 10 --no-cpu-baseline --fp32-steps 0 --two-video-steps 0 --no-pmc-traffic ${EXTRA:-} > $OUT/bench.log 2>&1
head -3 /tmp/tq/s_kernel_trace.csv > $OUT/head.txt
python3 - /tmp/tq/s_kernel_trace.csv $OUT/frame_queues.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(ev) if "disp_metrics_finish" in r["Kernel_Name"]]
a, b = marks[-3], marks[-2]
fr = ev[a + 1:b + 1]
t0 = int(fr[0]["Start_Timestamp"])
with open(sys.argv[2], "w") as f:
    for r in fr:
        f.write("%9.1f us %7.1f us  queue %-4s stream %-4s %s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                r.get("Queue_Id", "?"), r.get("Stream_Id", "?"), r["Kernel_Name"][:60]))
PY
tail -1 $OUT/bench.log | cut -c1-120; cat $OUT/head.txt | cut -c1-400
