"""per-stream busy time and a few key kernel durations over the last steps of a bench run (rocpd database)"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
ks = c.execute("select start,end,name,stream_id,queue_id from kernels order by start").fetchall()
g = [k for k in ks if "gcc_fwd" in k[2]]
t0, t1 = g[-12][0], g[-2][0]   # 5 steps (two gcc_fwd per step)
seg = [k for k in ks if t0 <= k[0] < t1]
print("window ms/step", (t1 - t0) / 5e6)
import collections
busy = collections.Counter(); n = collections.Counter()
for s, e, nm, sid, qid in seg:
    busy[(sid, qid)] += e - s; n[(sid, qid)] += 1
for k in sorted(busy): print("stream,queue", k, f"busy {busy[k]/5e6:7.3f} ms/step  kernels/step {n[k]/5:6.1f}")
for key in ("fps_kernel<48", "fps_kernel<12", "kg_query", "gcc_fwd", "gcc_bwd_data", "contrast_backward_kernel<32", "nn3_grid"):
    d = [(e - s) / 1e3 for s, e, nm, sid, qid in seg if key in nm]
    if d: print(f"{key:30s} n={len(d):3d} avg {sum(d)/len(d):9.1f} us  max {max(d):9.1f}")
