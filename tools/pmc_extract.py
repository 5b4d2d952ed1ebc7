"""Summarise a rocprofv3 --pmc pass (rocpd sqlite) into per-kernel sums over the LAST nsteps train steps (the window between
launches of a once-per-step marker kernel; warm-up steps -- MIOpen find mode, allocator growth -- lie before it):
python pmc_extract.py db counter out.csv [nsteps=3] [marker=ce_forward_kernel]"""
import csv, sqlite3, sys
db, counter, out = sys.argv[1:4]
nsteps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
marker = sys.argv[5] if len(sys.argv) > 5 else "ce_forward_kernel"
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
view = "counters_collection" if "counters_collection" in tabs else None
if view is None:
    print("tables:", tabs); sys.exit(1)
cols = [d[0] for d in c.execute(f"select * from {view} limit 1").description]
namecol = "kernel_name" if "kernel_name" in cols else "name"
order = next((x for x in ("dispatch_id", "start", "id") if x in cols), None)
rows = c.execute(f"select {namecol}, {order}, value from {view} where counter_name=? order by {order}", (counter,)).fetchall()
marks = [r[1] for r in rows if marker in r[0]]
lo, hi = (marks[-nsteps - 1], marks[-1]) if len(marks) > nsteps else (rows[0][1], rows[-1][1] + 1)
agg = {}
for name, o, v in rows:
    if lo <= o < hi:
        d = agg.setdefault(name, [0, 0.0])
        d[0] += 1; d[1] += v
res = sorted(agg.items(), key=lambda kv: -kv[1][1])
with open(out, "w") as f:
    w = csv.writer(f); w.writerow(["kernel", "counter", "dispatches", "avg_per_dispatch", "sum"])
    for k, (n, s) in res: w.writerow([k[:160], counter, n, round(s / n, 3), round(s, 3)])
print(f"{counter}: window of {nsteps} steps ({len(marks)} marker launches seen, ordered by {order}); top:")
for k, (n, s) in res[:8]: print(f"{s / n:14.1f} avg x{n:5d}  {k[:90]}")
