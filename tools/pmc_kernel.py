"""Average rocprofv3 --pmc counters per kernel over one or more counter_collection CSVs:
   python tools/pmc_kernel.py <kernel substring> <csv> [<csv> ...]   -> JSON {counter: mean per launch, ...}"""
import csv
import json
import sys
from collections import defaultdict

acc = defaultdict(list)
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        if sys.argv[1] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, v in acc.items():
    v = v[len(v) // 4:]  # skip warm-up launches
    out[k] = sum(v) / len(v)
out["_launches_averaged"] = {k: len(v) - len(v) // 4 for k, v in acc.items()}
print(json.dumps(out, indent=1))
