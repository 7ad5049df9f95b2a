#!/bin/bash
# usage: pmc_conv.sh cin cout k H W dil   (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"; do
rm -rf /tmp/pm; rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pm -o c -- python $GRAFT_REPO_ROOT/tools/time_one_conv.py "$@" > /dev/null 2>&1
python3 -c "
import csv,collections
rows=list(csv.DictReader(open('/tmp/pm/c_counter_collection.csv')))
d=collections.defaultdict(float); n=collections.defaultdict(int)
for r in rows:
    if 'conv_mfma' in r['Kernel_Name']:
        d[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
print(' '.join(f'{k}={d[k]/n[k]:.3g}' for k in d))
kn=[r for r in rows if 'conv_mfma' in r['Kernel_Name']][-1]
print(kn['Kernel_Name'][:50], 'vgpr', kn.get('VGPR_Count'), 'lds', kn.get('LDS_Block_Size'), 'grid', kn.get('Grid_Size'))
"
done
