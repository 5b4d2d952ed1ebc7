cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/profpw
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/profpw -o r -- python3 scratch/pw_bench.py > gpurun_out/pw_bench_x.log 2>&1
python3 - <<PY
import sqlite3
c=sqlite3.connect("/tmp/profpw/r_results.db")
rows=c.execute("select name,grid_x,grid_y,grid_z,count(*),avg(end-start)/1e3,min(start) from kernels where name like '%gm_gemm%' or name like '%pw_%' group by name,grid_x,grid_y,grid_z having count(*)>=10 order by min(start)").fetchall()
for r in rows: print(f"{r[5]:8.1f} us x{r[4]:3d} grid=({r[1]},{r[2]},{r[3]}) {r[0][10:60]}")
PY
