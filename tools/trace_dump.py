"""Condense a rocprofv3 kernel_trace.csv: start(us), duration(us), queue, short kernel name; last N rows."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
with open(sys.argv[2], "w") as f:
    for r in rows:
        name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:48]
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        f.write(f"{(st - t0) / 1e3:.1f},{(en - st) / 1e3:.1f},{r.get('Queue_Id', '')},{name}\n")
