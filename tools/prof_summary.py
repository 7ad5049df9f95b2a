"""Summarise a rocprofv3 kernel_stats.csv: python tools/prof_summary.py <csv> [frames]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
frames = float(sys.argv[2]) if len(sys.argv) > 2 else 25.0
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms = {tot/frames/1e6:.2f} ms/frame over {frames:.0f} frames")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print(f"{r['Name'][:64]:64s} {r['Calls']:>6s} {int(r['TotalDurationNs'])/frames/1e6:8.3f} ms/frame {float(r['AverageNs'])/1e3:8.1f} us")
