#!/bin/bash
# Run on the GPU box: rocprofv3 kernel statistics of bench.py (serial streams and default schedule) and the
# PMC passes (HBM traffic, MFMA busy) -> gpurun_out/prof_${ROUND}${TAG}/.  tools/make_profile_md.py turns them into
# the committed summaries under profiles/.
#   ROUND=r03 TAG=_cfg4_1280x384 ARGS="--height 384 --width 1280" tools/profile_round.sh      (BASELINE.json configs[3])
#   ROUND=r03 TAG=_cfg5_bf16_640x512 ARGS="--precision bf16 --height 512 --width 640" ...      (configs[4])
ROUND=${ROUND:-r03}; TAG=${TAG:-}; ARGS=${ARGS:-}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_${ROUND}${TAG}; mkdir -p $OUT
cp $R/codd_amd/tuned/mi355x.json $OUT/tune_db.json   # the shipped launch configurations (what bench.py runs by default)
python $R/bench.py $ARGS --no-cpu-baseline --fp32-steps 0 --two-video-steps 0 --no-pmc-traffic --tune-db $OUT/tune_db.json > $OUT/bench_plain.log 2>&1   # un-profiled reference run
for mode in serial default; do
  flag=""; [ $mode = serial ] && flag="--serial-streams"
  rm -rf /tmp/st_$mode
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$mode -o s -- python $R/bench.py $ARGS --steps 60 --no-cpu-baseline --fp32-steps 0 --two-video-steps 0 --no-pmc-traffic --tune-db $OUT/tune_db.json $flag > $OUT/bench_$mode.log 2>&1
  cp /tmp/st_$mode/s_kernel_stats.csv $OUT/${mode}_kernel_stats.csv
  [ $mode = default ] && python3 $R/tools/timeline_gaps.py /tmp/st_$mode/s_kernel_trace.csv --dump $OUT/frame_sequence.txt > $OUT/timeline_default.txt 2>&1
done
CMD="python $R/bench.py $ARGS --no-cpu-baseline --fp32-steps 0 --two-video-steps 0 --no-pmc-traffic --tune-db $OUT/tune_db.json --serial-streams --steps 4 --prewarm 2 --warmup 1 --no-graph"
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/pm_$tag
  timeout 280 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pm_$tag -o p -- $CMD > /dev/null 2>&1
  python3 - $tag /tmp/pm_$tag/p_counter_collection.csv $OUT/pmc_$tag.csv <<'PY'
import csv, sys, collections
tag, src, dst = sys.argv[1:4]
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(src)):
    k = (r["Kernel_Name"], r["Counter_Name"])
    acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
with open(dst, "w") as f:
    w = csv.writer(f); w.writerow(["kernel", "counter", "sum", "launches"])
    for (kn, cn), (s, n) in sorted(acc.items()):
        w.writerow([kn, cn, s, n])
PY
done
tail -1 $OUT/bench_plain.log | cut -c1-200; ls -la $OUT
