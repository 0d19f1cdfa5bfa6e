#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (.db) kernel trace: per-kernel stats CSV (like --stats) plus, for the
LAST `--last` dispatches of the timed region, the sum of kernel durations vs the wall span (gaps)."""
import argparse, csv, re, sqlite3, sys

ap = argparse.ArgumentParser()
ap.add_argument("db"); ap.add_argument("--csv"); ap.add_argument("--top", type=int, default=30)
a = ap.parse_args()
db = sqlite3.connect(a.db); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
names = {r[0]: r[1] for r in cur.execute(f"select id, kernel_name from {ks}")}
rows = list(cur.execute(f"select kernel_id, start, end from {kd} order by start"))
stat = {}
for kid, s, e in rows:
    st = stat.setdefault(names[kid], [0, 0, 1 << 62, 0])
    st[0] += 1; st[1] += e - s; st[2] = min(st[2], e - s); st[3] = max(st[3], e - s)
tot = sum(v[1] for v in stat.values())
out = sorted(stat.items(), key=lambda kv: -kv[1][1])
if a.csv:
    with open(a.csv, "w", newline="") as f:
        w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for n, v in out: w.writerow([n, v[0], v[1], "%.1f" % (v[1] / v[0]), "%.2f" % (100 * v[1] / tot), v[2], v[3]])
short = lambda n: re.sub(r"\(anonymous namespace\)::|void |\(.*", "", n)[:60]
for n, v in out[:a.top]:
    print("%-60s calls %5d total %9.3f ms avg %8.1f us  %5.2f%%" % (short(n), v[0], v[1] / 1e6, v[1] / v[0] / 1e3, 100 * v[1] / tot))
span = rows[-1][2] - rows[0][1]
print("all kernels: busy %.2f ms, span %.2f ms" % (tot / 1e6, span / 1e6))
