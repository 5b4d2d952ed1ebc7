"""Summarise a rocprofv3 --pmc pass (rocpd sqlite) into per-kernel averages: python pmc_extract.py db counter out.csv"""
import csv, sqlite3, sys
db, counter, out = sys.argv[1:4]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
view = "counters_collection" if "counters_collection" in tabs else None
if view is None:
    print("tables:", tabs); sys.exit(1)
cur = c.execute(f"select * from {view} limit 1")
cols = [d[0] for d in cur.description]
print(cols)
namecol = "kernel_name" if "kernel_name" in cols else "name"
rows = c.execute(f"select {namecol}, counter_name, count(*), avg(value), sum(value) from {view} where counter_name=? group by {namecol} order by 5 desc", (counter,)).fetchall()
with open(out, "w") as f:
    w = csv.writer(f); w.writerow(["kernel", "counter", "dispatches", "avg_per_dispatch", "sum"])
    for r in rows: w.writerow([r[0][:160], r[1], r[2], round(r[3], 3), round(r[4], 3)])
for r in rows[:12]: print(f"{r[3]:14.1f} avg x{r[2]:5d}  {r[0][:90]}")
