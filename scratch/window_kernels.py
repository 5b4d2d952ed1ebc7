"""top kernels inside a steady-state window of a bench.py trace: between the i-th and j-th launch of the first-level
FPS kernel (one per step): python window_kernels.py db i j [top]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1]); i, j = int(sys.argv[2]), int(sys.argv[3]); top = int(sys.argv[4]) if len(sys.argv) > 4 else 30
f = c.execute("select start from kernels where name like '%fps_kernel<48%' or name like '%fps_kernel_l2%' order by start").fetchall()
t0, t1 = f[i][0], f[j][0]; steps = j - i
rows = c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3 from kernels where start>=? and start<? group by name order by 3 desc", (t0, t1)).fetchall()
print(f"window {(t1-t0)/1e6/steps:.2f} ms/step over {steps} steps; kernel time {sum(r[2] for r in rows)/steps/1e3:.2f} ms/step; kernels/step {sum(r[1] for r in rows)/steps:.0f}")
for r in rows[:top]: print(f"{r[2]/steps:8.1f} us/step {r[1]/steps:6.1f}x avg {r[3]:8.1f}  {r[0][:100]}")
