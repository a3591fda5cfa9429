"""Average a rocprofv3 --pmc counter over the launches of one kernel: python tools/pmc_parse.py <csv> <kernel substring>"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
acc = defaultdict(list)
for r in rows:
    if sys.argv[2] in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    v = v[len(v) // 2:]  # skip warm-up launches
    print(k, sum(v) / len(v), len(v))
