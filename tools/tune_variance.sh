#!/bin/bash
# dev aid: run bench.py several times with autotune and list, per run, fps + the tuned configuration of the big layers
for i in 1 2 3 4 5; do
  CODD_BENCH_VERBOSE=1 python bench.py --no-cpu-baseline > /tmp/tv_out.txt 2> /tmp/tv_err.txt
  python3 - <<'PY'
import json, re
d = json.loads(open('/tmp/tv_out.txt').read().strip().splitlines()[-1])
print("FPS", d["value"], d["roofline"]["frac"])
for l in open('/tmp/tv_err.txt'):
    m = re.search(r"autotune (.*?) heuristic (\(.*?\)) ([\d.]+) us -> (\(.*?\)) ([\d.]+) us", l)
    if m and float(m.group(3)) > 45:
        print("   %-36s %s %6.1f -> %s %6.1f" % m.groups())
PY
done
