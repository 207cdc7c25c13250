"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into a per-kernel CSV: calls, total/avg/min/max us, %.
usage: python tools/rocpd_stats.py <results.db> <out.csv>"""
import csv
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
rows = c.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                 f"from kernels group by {name_col} order by sum(end-start) desc").fetchall()
tot = sum(r[2] for r in rows) or 1
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "MinUs", "MaxUs", "Percentage"])
    for r in rows:
        w.writerow([r[0], r[1], round(r[2] / 1e3, 3), round(r[3] / 1e3, 3), round(r[4] / 1e3, 3), round(r[5] / 1e3, 3),
                    round(100.0 * r[2] / tot, 3)])
print("wrote", out, len(rows), "kernels")
