"""every launch of the kernels whose name contains <pattern> in the last step of a rocprofv3 rocpd database, in launch order,
with grid / workgroup size and duration:  python kernel_calls.py db pattern [marker=ce_forward_kernel]"""
import sqlite3, sys
db, pat = sys.argv[1], sys.argv[2]
marker = sys.argv[3] if len(sys.argv) > 3 else "ce_forward_kernel"
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
want = [k for k in ("grid_x", "grid_y", "grid_z", "workgroup_x", "grid_size_x", "grid_size_y", "grid_size_z", "workgroup_size_x", "lds_size", "lds_block_size") if k in cols]
ks = c.execute(f"select start,end,name,{','.join(want) if want else '0'} from kernels order by start").fetchall()
g = [k[0] for k in ks if marker in k[2]]
t0, t1 = g[-2], g[-1]
print("columns:", cols)
for k in ks:
    if t0 <= k[0] < t1 and pat in k[2]:
        print(f"{(k[0]-t0)/1e3:9.1f} us  dur {(k[1]-k[0])/1e3:7.1f} us  {dict(zip(want, k[3:]))}  {k[2][:70]}")
