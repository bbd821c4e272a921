#!/usr/bin/env python3
"""Per-kernel table from a rocprofv3 rocpd database (<dir>/*_results.db): calls, avg/min us, total ms, share."""
import glob, sqlite3, sys
d = sys.argv[1]; top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
f = d if d.endswith(".db") else glob.glob(d + "/**/*_results.db", recursive=True)[0]
c = sqlite3.connect(f)
rows = c.execute("select name, count(*), avg(end-start), min(end-start), sum(end-start) from kernels group by name order by 5 desc").fetchall()
tot = sum(r[4] for r in rows)
print("Name,Calls,AverageNs,MinNs,TotalDurationNs,Percentage")
for r in rows[:top]:
    n = r[0].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f'"{n[:90]}",{r[1]},{r[2]:.0f},{r[3]},{r[4]},{100 * r[4] / tot:.2f}')
print(f'"TOTAL",{sum(r[1] for r in rows)},,,{tot},100.0')
