"""Dev aid: from a rocprofv3 kernel trace csv, take the last complete frame (between two consecutive
disp_metrics_finish kernels) and report busy time, idle gaps and concurrency."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
              "%sx%sx%s/%s" % (r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), r.get("Grid_Size_Z", "?"), r.get("Workgroup_Size_X", "?")),
              r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows), key=lambda t: t[0])
marks = [i for i, e in enumerate(ev) if "disp_metrics_finish" in e[2]]
a, b = marks[-3], marks[-2]
fr = ev[a + 1:b + 1]
if len(sys.argv) > 3 and sys.argv[2] == "--dump":  # the frame's launches in start order: offset, duration, queue, name, grid
    with open(sys.argv[3], "w") as f:
        for s_, e_, n_, g_, q_ in fr:
            f.write("%9.1f us  %7.1f us  q%-4s %-70s %s\n" % ((s_ - fr[0][0]) / 1e3, (e_ - s_) / 1e3, q_, n_[:70], g_))
fr = [e[:3] for e in fr]
t0, t1 = fr[0][0], max(e[1] for e in fr)
print("frame: %d kernels, wall %.2f ms, sum of kernel time %.2f ms" % (len(fr), (t1 - t0) / 1e6, sum(e[1] - e[0] for e in fr) / 1e6))
# union coverage and gaps
cur_end, busy, gaps = fr[0][0], 0, []
last = None
for s, e, n in fr:
    if s > cur_end:
        gaps.append((s - cur_end, last, n)); cur_end = s
    if e > cur_end:
        busy += e - max(s, cur_end); cur_end = e; last = n
print("busy (>=1 kernel running) %.2f ms, idle %.2f ms in %d gaps" % (busy / 1e6, sum(g[0] for g in gaps) / 1e6, len(gaps)))
import collections
c = collections.Counter()
for g, p, n in gaps:
    c[(p[:28], n[:28])] += g
for (p, n), g in c.most_common(14):
    print("  idle %.3f ms  after %-28s before %s" % (g / 1e6, p, n))
# time with exactly one kernel running, by kernel
pts = []
for s, e, n in fr:
    pts.append((s, 1, n)); pts.append((e, -1, n))
pts.sort(key=lambda t: (t[0], t[1]))
active, solo, prev, conc = {}, collections.Counter(), None, collections.Counter()
for t, d, n in pts:
    if prev is not None and active:
        k = sum(active.values())
        conc[min(k, 4)] += t - prev
        if k == 1:
            solo[[m for m, v in active.items() if v][0][:60]] += t - prev
    active[n] = active.get(n, 0) + d
    if active[n] == 0:
        del active[n]
    prev = t
print("concurrency histogram (ms): " + ", ".join("%s: %.2f" % (("%d" % k if k < 4 else "4+"), v / 1e6) for k, v in sorted(conc.items())))
print("time with exactly ONE kernel on the device, by kernel:")
for n, v in solo.most_common(14):
    print("  %.3f ms  %s" % (v / 1e6, n))
