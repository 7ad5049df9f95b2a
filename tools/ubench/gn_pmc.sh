#!/bin/bash
# dev (GPU box): SQ counters of the two Gauss-Newton builders in the stand-alone harness (gn_ablate_base.bin)
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/tools/ubench/gn_ablate_base.bin
for v in 0 1; do for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_LDS"; do
  rm -rf /tmp/pg
  GN3=$v CODD_GN_Q4=${Q4:-128} timeout 120 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pg -o p -- $B > /dev/null 2>&1
  python3 - /tmp/pg/p_counter_collection.csv $v <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if "se3_gn_build" in r["Kernel_Name"]:
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
print("pair" if sys.argv[2] == "1" else "J-entry", {k: round(v[0] / v[1]) for k, v in acc.items()})
PY
done; done
