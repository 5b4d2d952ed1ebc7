"""non-amc kernels of the steady-state step per stream (rocpd db): the torch / library glue that is left"""
import collections, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
ks = c.execute("select start,end,name,stream_id from kernels order by start").fetchall()
g = [k for k in ks if "gcc_fwd" in k[2]]
# bench.py ends with 5 lone feature-graph replays and 3 eager steps (16 gcc_fwd launches): step back into the timed loop
t0, t1 = g[-16 - 11][0], g[-16 - 1][0]; nsteps = 5
seg = [k for k in ks if t0 <= k[0] < t1]
tot = collections.Counter(); cnt = collections.Counter(); allt = collections.Counter()
for s, e, nm, sid in seg:
    allt[sid] += e - s
    if "amc::" in nm: continue
    tot[(sid, nm[:110])] += e - s; cnt[(sid, nm[:110])] += 1
print("busy us/step per stream:", {k: round(v / nsteps / 1e3) for k, v in allt.items()})
print("glue us/step per stream:", {sid: round(sum(v for (s, _), v in tot.items() if s == sid) / nsteps / 1e3) for sid in allt})
for (sid, nm), v in sorted(tot.items(), key=lambda kv: -kv[1])[:40]:
    print(f"s{sid} {v/nsteps/1e3:7.1f} us/step {cnt[(sid,nm)]/nsteps:6.1f}x  {nm}")
