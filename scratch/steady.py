"""steady-state kernel statistics of the bench's main stream from a rocprofv3 rocpd database"""
import collections, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select stream_id,count(*) from kernels group by stream_id").fetchall(); print(rows)
sid = max(rows, key=lambda r: r[1])[0]
ks = c.execute("select start,end,name from kernels where stream_id=? order by start", (sid,)).fetchall()
idx = [i for i, k in enumerate(ks) if "gcc_fwd" in k[2]]
a, b = idx[-12], idx[-2]
seg = ks[a:b]
n = len(seg); busy = sum(e - s for s, e, _ in seg); span = seg[-1][1] - seg[0][0]
print("kernels/step", n / 5, "busy ms/step", busy / 5e6, "span ms/step", span / 5e6)
small = collections.Counter(); st = collections.Counter()
for s, e, nm in seg:
    if e - s < 6000: small[nm[:90]] += 1; st[nm[:90]] += (e - s)
for k, v in small.most_common(30): print(f"{v/5:6.1f}x {st[k]/5e3:7.1f}us  {k}")
print("small kernels/step", sum(small.values()) / 5, "us", sum(st.values()) / 5e3)
