"""Per-kernel summary of a rocprofv3 --kernel-trace results database: python tools/kstats.py <results.db> [top]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
q = (f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3, sum(d.end-d.start)/1e6 from {kd} d join {ks} s "
     f"on d.kernel_id=s.id group by s.kernel_name order by 4 desc limit {top}")
tot = list(db.execute(f"select sum(end-start)/1e6, count(*) from {kd}"))[0]
print(f"total {tot[0]:.2f} ms in {tot[1]} dispatches")
for name, calls, avg, total in db.execute(q):
    print(f"{calls:6d} {avg:10.1f} us {total:9.2f} ms  {name[:120]}")
