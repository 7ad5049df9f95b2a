#!/bin/bash
# dev (GPU box): wave-cycle accounting of the Gauss-Newton builder inside the frame (serial streams, eager):
# WAIT_ANY (parked on s_waitcnt / barrier) + WAIT_INST_ANY (issue stall: dependency / pipe) + ACTIVE_INST_ANY ~ WAVE_CYCLES
# (MI355X_MICROARCH.md, rocprofv3 PMC slots), VALU / scalar shares of the active cycles, instruction counts per launch.
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SMEM"; do
  rm -rf /tmp/pg
  timeout 280 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pg -o p -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --fp32-steps 0 --two-video-steps 0 --no-pmc-traffic --serial-streams --steps 2 --prewarm 1 --warmup 1 --no-graph > /dev/null 2>&1
  python3 - /tmp/pg/p_counter_collection.csv <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
try:
    rows = list(csv.DictReader(open(sys.argv[1])))
except Exception as e:
    print("no counters:", e); rows = []
for r in rows:
    if "se3_gn_build" in r["Kernel_Name"]:
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
print({k: round(v[0] / v[1]) for k, v in acc.items()}, "launches", max([v[1] for v in acc.values()] or [0]))
PY
done
