#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pm; rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d /tmp/pm -o c -- python $GRAFT_REPO_ROOT/tools/time_one_conv.py "$@" > /dev/null 2>&1
python3 -c "
import csv,collections
rows=list(csv.DictReader(open('/tmp/pm/c_counter_collection.csv')))
tr={r['Dispatch_Id']:r for r in csv.DictReader(open('/tmp/pm/c_kernel_trace.csv'))}
for r in rows:
    if 'conv_mfma' in r['Kernel_Name'] and r['Counter_Name']=='GRBM_GUI_ACTIVE':
        t=tr[r['Dispatch_Id']]; dur=int(t['End_Timestamp'])-int(t['Start_Timestamp'])
        print('GUI_ACTIVE', r['Counter_Value'], 'dur_ns', dur, 'GHz', float(r['Counter_Value'])/dur)
"
