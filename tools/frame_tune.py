"""Dev aid: in-context check of the launch configuration of the heavy GRU layers.  For each signature, every
alternative is written into a copy of the shipped tune db and the whole-frame rate is measured with bench.py
(a fresh process each: graph re-captured).  usage (GPU box): python tools/frame_tune.py [steps]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DB = os.path.join(ROOT, "codd_amd", "tuned", "mi355x.json")
steps = sys.argv[1] if len(sys.argv) > 1 else "40"
ALTS = {
    "768,128,3,3,4,0|72,120,1,1,1,1,1,1,0": [[1, 4, 8, 4], [4, 4, 16, 4], [1, 8, 16, 4], [2, 4, 8, 2], [1, 9, 8, 4]],
    "256,128,3,3,4,0|72,120,1,1,1,1,1,1,0": [[1, 9, 16, 4], [1, 9, 12, 4], [1, 4, 8, 2], [1, 4, 12, 4]],
    "256,128,3,3,4,0|72,120,1,1,1,4,4,4,0": [[1, 4, 8, 2], [1, 9, 12, 4], [1, 9, 16, 4], [1, 4, 12, 4]],
    "256,196,3,3,4,0|72,120,1,1,1,1,1,1,0": [[1, 4, 8, 2], [1, 9, 16, 4], [1, 4, 12, 4]],
    "256,256,3,3,4,0|72,120,1,1,1,1,1,1,0": [[1, 9, 16, 4], [1, 4, 8, 2], [1, 4, 12, 4]],
    "128,128,3,3,2,0|72,120,1,1,1,1,1,1,0": [[1, 9, 16, 2], [1, 4, 12, 2], [1, 4, 8, 2]],
    "128,128,3,3,2,0|72,120,1,1,1,4,4,4,0": [[1, 9, 16, 2], [1, 4, 12, 2], [1, 4, 8, 2]],
}


def fps(db):
    path = "/tmp/frame_tune_db.json"
    json.dump(db, open(path, "w"))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--tune-db", path, "--steps", steps,
                          "--prewarm", "60"], capture_output=True, text=True).stdout.strip().splitlines()[-1]
    return json.loads(out)["value"]


db = json.load(open(DB))
fps(db)  # box warm-up
base = max(fps(db), fps(db))
print("shipped db: %.2f frames/s" % base, flush=True)
for sig, alts in ALTS.items():
    best, best_f = db[sig], base
    for a in alts:
        if a == db[sig]:
            continue
        trial = dict(db); trial[sig] = a
        f = fps(trial)
        print("  %-44s %s -> %s : %.2f" % (sig, db[sig], a, f), flush=True)
        if f > best_f + 0.15:
            best, best_f = a, f
    if best != db[sig]:
        db[sig] = best
        base = best_f
        print("  keep %s for %s (%.2f)" % (best, sig, best_f), flush=True)
json.dump(db, open(os.path.join(ROOT, "gpurun_out", "frame_tuned_db.json"), "w"), indent=0, sort_keys=True)
print("final %.2f frames/s -> gpurun_out/frame_tuned_db.json" % fps(db))
