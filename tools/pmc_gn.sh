#!/bin/bash
# dev (GPU box): SQ counters of the Gauss-Newton builder (both variants) -> what bounds it
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  rm -rf /tmp/pg$v
  CODD_GN_MFMA=$v timeout 280 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d /tmp/pg$v -o p -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --fp32-steps 0 --two-video-steps 0 --no-pmc-traffic --serial-streams --steps 2 --prewarm 1 --warmup 1 --no-graph > /dev/null 2>&1
  python3 - /tmp/pg$v/p_counter_collection.csv <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if "se3_gn_build" in r["Kernel_Name"]:
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
print({k: round(v[0] / v[1]) for k, v in acc.items()})
PY
done
