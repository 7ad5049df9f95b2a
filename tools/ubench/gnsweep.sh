cd /tmp && export TMPDIR=/tmp
for v in 0 1 2; do for q in 64 96 128 160; do
CODD_GN_VAR=$v CODD_GN_Q4=$q rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gn_${v}_$q -o g -- python $GRAFT_REPO_ROOT/tools/time_ops.py gn32 > /dev/null 2>&1
python3 -c "
import csv
o=[]
for r in list(csv.DictReader(open('/tmp/gn_${v}_$q/g_kernel_stats.csv')))[:3]: o.append('%s %.1f' % (r['Name'][7:12], float(r['AverageNs'])/1e3))
print('var $v q4 $q:', ' | '.join(o))
"
done; done
