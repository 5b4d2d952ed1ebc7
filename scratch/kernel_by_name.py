"""per-launch-shape durations of kernels whose name contains argv[2] in a rocprofv3 rocpd database"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name,grid_x,grid_y,grid_z,count(*),avg(end-start)/1e3,vgpr_count,lds_size from kernels where name like ? group by name,grid_x,grid_y,grid_z order by name, 6 desc", ('%' + sys.argv[2] + '%',)).fetchall()
for r in rows: print(f"{r[5]:8.1f} us x{r[4]:3d} grid=({r[1]},{r[2]},{r[3]}) vgpr={r[6]} lds={r[7]} {r[0][:70]}")
