"""Summarise rocprofv3 counter_collection.csv + kernel_trace.csv of one run: per kernel avg counter value and avg duration."""
import csv, collections, sys, json
d = sys.argv[1]; prefix = sys.argv[2] if len(sys.argv) > 2 else "run"
dur = collections.defaultdict(list)
for r in csv.DictReader(open(f"{d}/{prefix}_kernel_trace.csv")):
    dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f"{d}/{prefix}_counter_collection.csv")):
    cnt[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = []
for k, cs in cnt.items():
    if "selftok" not in k and "Cijk" not in k:
        continue
    row = {"kernel": k[:60], "calls": len(dur.get(k, [])), "avg_us": round(sum(dur[k]) / max(1, len(dur[k])) / 1e3, 1) if k in dur else None}
    for c, v in cs.items():
        row[c] = round(sum(v) / len(v), 1)
    out.append(row)
out.sort(key=lambda r: -(r["avg_us"] or 0) * r["calls"])
for r in out[:12]:
    print(json.dumps(r))
