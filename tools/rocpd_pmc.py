#!/usr/bin/env python3
"""Print per-kernel PMC counter averages from a rocprofv3 results database (rocpd sqlite)."""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
try:
    rows = cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
except Exception as e:
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    print("counters_collection columns:", cols); raise
print("kernel,counter,avg_per_dispatch,dispatches")
for k, c, v, n in rows:
    print("%s,%s,%.1f,%d" % (k.split("(")[0].replace(",", ";"), c, v, n))
