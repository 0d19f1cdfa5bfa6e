#!/usr/bin/env python3
"""The last launches of one step's backward in a rocprofv3 rocpd kernel trace, per queue, with start / end relative to the end of the step:
what runs beside the first convolution's own chain (reduce -> weight gradient -> fold -> unpack) at the tail.
    python tools/tail_timeline.py trace.db [n_last]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 28
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]; ks = [t for t in tabs if "kernel_symbol" in t][0]
names = {r[0]: r[1] for r in cur.execute(f"select id, kernel_name from {ks}")}
short = lambda n: re.sub(r"\(anonymous namespace\)::|void |\(.*", "", n)[:60]
rows = [(s, e, q, short(names[k])) for k, q, s, e in cur.execute(f"select kernel_id, queue_id, start, end from {kd} order by start")]
marks = [i for i, r in enumerate(rows) if "pack_weights" in r[3]]
lo, hi = marks[-2], marks[-1]
step = rows[lo:hi]
t1 = max(r[1] for r in step)
# the step's own end = the unpack launch
up = [r for r in step if "unpack_wgrads" in r[3]]
tend = up[-1][1] if up else t1
print("step span %.3f ms; launches ending within the last 600 us before the end of unpack_wgrads:" % ((tend - step[0][0]) / 1e6))
for r in step:
    if r[1] > tend - 600e3 and r[0] <= tend:
        print("  q%-3d %8.1f .. %8.1f us  (%6.1f us)  %s" % (r[2], (r[0] - tend) / 1e3, (r[1] - tend) / 1e3, (r[1] - r[0]) / 1e3, r[3]))
