#!/usr/bin/env python3
"""Summarise a rocprofv3 results database (rocpd sqlite, the default output of rocprofv3 7.2
--kernel-trace --stats) as CSV: one row per kernel with calls, total/avg/min/max duration (us),
grid, workgroup, LDS, VGPR/SGPR counts.  Usage: rocpd_summary.py results.db > summary.csv"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
rows = cur.execute(
    "select name, count(*), sum(duration)/1e3, avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3, "
    "max(grid_x), max(workgroup_x), max(lds_size), max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(scratch_size) "
    "from kernels group by name order by sum(duration) desc").fetchall()
total = sum(r[2] for r in rows) or 1.0
print("kernel,calls,total_us,avg_us,min_us,max_us,percent,grid_x,workgroup_x,lds_bytes,vgpr,agpr,sgpr,scratch")
for r in rows:
    name = r[0].replace(",", ";")
    print("%s,%d,%.3f,%.3f,%.3f,%.3f,%.2f,%d,%d,%d,%d,%d,%d,%d" % (name, r[1], r[2], r[3], r[4], r[5], 100 * r[2] / total,
                                                                 r[6], r[7], r[8], r[9], r[10], r[11], r[12]))
