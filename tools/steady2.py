"""Per-kernel statistics of the steady-state train step from a rocprofv3 --kernel-trace rocpd database of
`bench.py --lean` (the process ends right after the timed loop):  python steady2.py db out.csv [nsteps=5] [marker=ce_forward_kernel]
The window is the last nsteps steps, delimited by launches of a kernel that runs exactly once per step."""
import collections, csv, sqlite3, sys
db, out = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 5
marker = sys.argv[4] if len(sys.argv) > 4 else "ce_forward_kernel"
c = sqlite3.connect(db)
ks = c.execute("select start,end,name,stream_id from kernels order by start").fetchall()
g = [k for k in ks if marker in k[2]]
t0, t1 = g[-n - 1][0], g[-1][0]
seg = [k for k in ks if t0 <= k[0] < t1]
agg = collections.defaultdict(list)
for s, e, nm, sid in seg: agg[(nm, sid)].append((e - s) / 1e3)
tot = sum(sum(v) for v in agg.values())
rows = sorted(agg.items(), key=lambda kv: -sum(kv[1]))
with open(out, "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Stream", "CallsPerStep", "UsPerStep", "AverageUs", "MinUs", "MaxUs", "Percentage"])
    for (nm, sid), v in rows:
        w.writerow([nm, sid, round(len(v) / n, 2), round(sum(v) / n, 1), round(sum(v) / len(v), 2), round(min(v), 2), round(max(v), 2),
                    round(100 * sum(v) / tot, 2)])
busy = collections.Counter()
for (nm, sid), v in agg.items(): busy[sid] += sum(v)
print(f"window {(t1 - t0) / n / 1e6:.3f} ms/step over {n} steps; kernels/step {len(seg) / n:.0f}; busy us/step per stream "
      + str({k: round(v / n) for k, v in sorted(busy.items())}))
amc = collections.Counter()
for (nm, sid), v in agg.items(): amc[(sid, "amc::" in nm)] += sum(v)
print("us/step per (stream, is-amc-kernel): " + str({k: round(v / n) for k, v in sorted(amc.items())}))
for (nm, sid), v in rows[:60]: print(f"s{sid} {sum(v)/n:8.1f} us/step {len(v)/n:6.1f}x avg {sum(v)/len(v):8.1f}  {nm[:120]}")
