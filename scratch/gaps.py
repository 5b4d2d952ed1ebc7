"""largest idle gaps on the bench's main stream in the last steps (rocprofv3 rocpd database): what sits around them"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select stream_id,count(*) from kernels group by stream_id").fetchall(); print(rows)
sid = max(rows, key=lambda r: r[1])[0]
ks = c.execute("select start,end,name from kernels where stream_id=? order by start", (sid,)).fetchall()
idx = [i for i, k in enumerate(ks) if "gcc_fwd" in k[2]]
a, b = idx[-6], idx[-2]
seg = ks[a:b]
gaps = sorted(((seg[i + 1][0] - seg[i][1], i) for i in range(len(seg) - 1)), reverse=True)
print("span ms", (seg[-1][1] - seg[0][0]) / 1e6, "busy ms", sum(e - s for s, e, _ in seg) / 1e6)
for g, i in gaps[:25]:
    print(f"{g/1e3:8.1f} us after {seg[i][2][:70]!r} before {seg[i+1][2][:70]!r}")
names = set(k[2][:60] for k in ks if "ccl" in k[2].lower())
print(names)
