"""all launches longer than <min_us> of the last step of a rocprofv3 rocpd database, in launch order (name, grid in workgroups, us)"""
import sqlite3, sys
db, min_us = sys.argv[1], float(sys.argv[2])
marker = sys.argv[3] if len(sys.argv) > 3 else "ce_forward_kernel"
c = sqlite3.connect(db)
ks = c.execute("select start,end,name,grid_x,grid_y,grid_z,workgroup_x from kernels order by start").fetchall()
g = [k[0] for k in ks if marker in k[2]]
t0, t1 = g[-2], g[-1]
for s, e, nm, gx, gy, gz, wx in ks:
    if t0 <= s < t1 and (e - s) / 1e3 >= min_us:
        print(f"{(s-t0)/1e3:9.1f} us  dur {(e-s)/1e3:7.1f} us  wgs ({gx//max(wx,1)},{gy},{gz})  {nm[:95]}")
