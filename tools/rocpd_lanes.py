#!/usr/bin/env python3
"""Per-queue (launch lane) timeline of the LAST step in a rocprofv3 rocpd kernel trace:
busy time, idle gaps and the tail each queue leaves behind the main lane.  The step boundary is
found from the `pack_weights` launch that opens every forward plan.

    python tools/rocpd_lanes.py trace_results.db [--steps 3]
"""
import argparse, re, sqlite3

ap = argparse.ArgumentParser()
ap.add_argument("db"); ap.add_argument("--steps", type=int, default=3); ap.add_argument("--gaps", type=int, default=12)
a = ap.parse_args()
db = sqlite3.connect(a.db); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
names = {r[0]: r[1] for r in cur.execute(f"select id, kernel_name from {ks}")}
short = lambda n: re.sub(r"\(anonymous namespace\)::|void |\(.*", "", n)[:44]
rows = [(s, e, q, short(names[k])) for k, q, s, e in cur.execute(f"select kernel_id, queue_id, start, end from {kd} order by start")]
marks = [i for i, r in enumerate(rows) if "pack_weights" in r[3]]
if len(marks) < a.steps + 1:
    raise SystemExit("not enough steps in the trace")
for si in range(a.steps):
    lo, hi = marks[-a.steps - 1 + si], marks[-a.steps + si]
    step = rows[lo:hi]
    t0, t1 = step[0][0], max(r[1] for r in step)
    print("step %d: %d launches, span %.3f ms (next step starts %.3f ms after this one)" % (si, len(step), (t1 - t0) / 1e6, (rows[hi][0] - t0) / 1e6))
    queues = sorted({r[2] for r in step})
    for q in queues:
        ks_ = [r for r in step if r[2] == q]
        busy = sum(r[1] - r[0] for r in ks_)
        first, last = ks_[0][0], max(r[1] for r in ks_)
        gaps = [(ks_[i + 1][0] - ks_[i][1], ks_[i][3], ks_[i + 1][3]) for i in range(len(ks_) - 1)]
        pos = [g for g in gaps if g[0] > 0]
        print("  queue %d: %4d launches, busy %.3f ms, active %.3f..%.3f ms, idle inside %.3f ms (median gap %.1f us)" % (
            q, len(ks_), busy / 1e6, (first - t0) / 1e6, (last - t0) / 1e6, sum(g[0] for g in pos) / 1e6,
            sorted(g[0] for g in pos)[len(pos) // 2] / 1e3 if pos else 0.0))
        if si == a.steps - 1:
            for g in sorted(gaps, key=lambda g: -g[0])[:a.gaps]:
                print("      gap %7.1f us after %-44s before %s" % (g[0] / 1e3, g[1], g[2]))
    # union busy over all queues
    ev = sorted([(r[0], 1) for r in step] + [(r[1], -1) for r in step])
    depth, last_t, idle, conc = 0, t0, 0, 0
    for t, d in ev:
        if depth == 0: idle += t - last_t
        if depth >= 2: conc += t - last_t
        depth += d; last_t = t
    print("  GPU idle (no kernel at all) %.3f ms; >=2 kernels in flight %.3f ms" % (idle / 1e6, conc / 1e6))
