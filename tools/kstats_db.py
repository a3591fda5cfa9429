"""Per-kernel averages from a rocprofv3 results database (rocprofv3 --kernel-trace --stats -d DIR -o NAME writes
DIR/NAME_results.db when no csv output format is asked for).   python tools/kstats_db.py path/to/results.db [n]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
q = (f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3, sum(d.end-d.start)/1e6 from {kd} d join {ks} s "
     f"on d.kernel_id=s.id group by s.kernel_name order by 4 desc limit {int(sys.argv[2]) if len(sys.argv) > 2 else 12}")
for r in c.execute(q):
    print(f"{r[1]:6d} {r[2]:9.1f} us {r[3]:9.1f} ms  {r[0][:120]}")
