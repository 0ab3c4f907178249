#!/usr/bin/env python3
"""Timeline of a rocprofv3 results database (rocpd sqlite; --kernel-trace --memory-copy-trace): every kernel dispatch and
memory copy in start order with start / end relative to the first event of the window, for the events of `n` consecutive
host calls in the middle of the run (a call = from one `planarize` kernel of band 0 to the next).
Usage: rocpd_timeline.py results.db [first_event_index] [count]     (--schema: list views and columns)"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
if "--schema" in sys.argv:
    for name, kind in cur.execute("select name, type from sqlite_master where type in ('table','view') order by name").fetchall():
        cols = [r[1] for r in cur.execute("pragma table_info('%s')" % name).fetchall()]
        print(kind, name, cols)
    sys.exit(0)
ev = []
for r in cur.execute("select name, start, end, stream_id, queue_id from kernels").fetchall():
    ev.append((r[1], r[2], "K", r[0].split("(")[0].replace("void mtm::", "").replace("mtm::", "")[:60], r[3]))
try:
    for r in cur.execute("select name, start, end, size, stream_id from memory_copies").fetchall():
        ev.append((r[1], r[2], "C", "%s %d B" % (r[0], r[3]), r[4]))
except sqlite3.Error as e:
    print("# no memory_copies view:", e)
ev.sort()
i0 = int(sys.argv[2]) if len(sys.argv) > 2 else len(ev) // 2
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
w = ev[i0:i0 + n]
t0 = w[0][0]
print("start_us,end_us,dur_us,kind,stream,name")
for s, e, k, name, st in w:
    print("%.1f,%.1f,%.1f,%s,%s,%s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, k, st, name))
