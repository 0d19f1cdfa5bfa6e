#!/usr/bin/env python3
"""MFMA-pipe busy fraction per kernel family from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass:
busy = SQ_VALU_MFMA_BUSY_CYCLES (cycles summed over the 1024 SIMDs) / (1024 x kernel cycles), kernel cycles =
GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs; MI355X_MICROARCH.md "DVFS give-back").
    python tools/rocpd_mfma_busy.py pass.db out.json"""
import collections, json, re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
T = lambda k: [t for t in tabs if k in t][0]
kd, ks, pe, pi = T("kernel_dispatch"), T("kernel_symbol"), T("rocpd_pmc_event"), T("rocpd_info_pmc")
cname = {r[0]: r[1] for r in cur.execute(f"select id, name from {pi}")}
names = {r[0]: r[1] for r in cur.execute(f"select id, kernel_name from {ks}")}
vals = collections.defaultdict(lambda: collections.defaultdict(float))
for ev, pid, v in cur.execute(f"select event_id, pmc_id, value from {pe}"):
    vals[ev][cname[pid]] += v
fam = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for kid, ev, s, e in cur.execute(f"select kernel_id, event_id, start, end from {kd}"):
    n = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", names[kid])
    n = re.sub(r"(_kernel|I[a-zL].*|E[vP].*)$", "", re.split(r"I(?=[Lt])", n)[0])
    f = fam[n]; c = vals.get(ev, {})
    f[0] += 1; f[1] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0); f[2] += c.get("GRBM_GUI_ACTIVE", 0.0); f[3] += e - s
out = {}
for n, (cnt, mf, gui, dur) in sorted(fam.items(), key=lambda kv: -kv[1][3]):
    if mf <= 0 or gui <= 0: continue
    cyc = gui / 8.0
    out[n] = {"launches": cnt, "mfma_busy_frac": mf / (1024.0 * cyc), "avg_us": dur / cnt / 1e3, "clock_GHz": cyc / dur,
              # GUI_ACTIVE/8/duration reads high on dispatches shorter than ~0.3 ms: also normalise by wall time at the spec clock
              "mfma_busy_of_peak_at_2.4GHz": mf / (1024.0 * dur * 2.4)}
    print("%-28s x%5d  MFMA busy %.3f  avg %.1f us  clock %.2f GHz" % (n, cnt, out[n]["mfma_busy_frac"], out[n]["avg_us"], out[n]["clock_GHz"]))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
