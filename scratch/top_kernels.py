import sqlite3, sys, csv
c = sqlite3.connect(sys.argv[1])
nsteps = c.execute("select count(*) from kernels where name like '%gcc_fwd%'").fetchone()[0] / 2
rows = c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 from kernels where name not like '%naive_conv%' group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
if len(sys.argv) > 2:
    with open(sys.argv[2], 'w') as f:
        w = csv.writer(f); w.writerow(['Name', 'Calls', 'TotalDurationUs', 'AverageUs', 'MinUs', 'MaxUs', 'Percentage', 'CallsPerStep', 'UsPerStep'])
        for r in rows: w.writerow([r[0], r[1], round(r[2], 1), round(r[3], 2), round(r[4], 2), round(r[5], 2), round(100 * r[2] / tot, 2), round(r[1] / nsteps, 2), round(r[2] / nsteps, 1)])
print('feature-graph executions', nsteps, 'total kernel us/step', round(tot / nsteps))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 45]: print(f"{r[2]/nsteps:8.1f} us/step {r[1]/nsteps:6.1f}x avg {r[3]:8.1f}  {r[0][:105]}")
