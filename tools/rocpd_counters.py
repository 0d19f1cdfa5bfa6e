#!/usr/bin/env python3
"""Per-kernel averages of every PMC counter in a rocprofv3 --pmc rocpd .db:  python tools/rocpd_counters.py x.db [name filter]"""
import collections, re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
T = lambda k: [t for t in tabs if k in t][0]
kd, ks, pe, pi = T("kernel_dispatch"), T("kernel_symbol"), T("rocpd_pmc_event"), T("rocpd_info_pmc")
cname = {r[0]: r[1] for r in cur.execute(f"select id, name from {pi}")}
names = {r[0]: r[1] for r in cur.execute(f"select id, kernel_name from {ks}")}
vals = collections.defaultdict(lambda: collections.defaultdict(float))
for ev, pid, v in cur.execute(f"select event_id, pmc_id, value from {pe}"):
    vals[ev][cname[pid]] += v
agg = collections.defaultdict(lambda: [0, collections.defaultdict(float), 0.0])
for kid, ev, s, e in cur.execute(f"select kernel_id, event_id, start, end from {kd}"):
    n = re.sub(r"\(anonymous namespace\)::|void |\(.*", "", names[kid])[:70]
    if flt and not re.search(flt, n): continue
    a = agg[n]; a[0] += 1; a[2] += e - s
    for c, v in vals.get(ev, {}).items(): a[1][c] += v
for n, (cnt, cs, dur) in sorted(agg.items(), key=lambda kv: -kv[1][2])[:int(sys.argv[3]) if len(sys.argv) > 3 else 12]:
    print("%s  x%d  avg %.1f us" % (n, cnt, dur / cnt / 1e3))
    for c, v in sorted(cs.items()): print("     %-32s %14.0f" % (c, v / cnt))
