#!/usr/bin/env python3
"""Per-kernel-family HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; rocpd .db).
Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both derived metrics are in KiB;
on gfx950 FETCH_SIZE tallies 128-B requests of wide coalesced reads at 64 B -> doubled here."""
import collections, json, re, sqlite3, sys


def per_kernel(path, counter):
    db = sqlite3.connect(path); cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    T = lambda k: [t for t in tabs if k in t][0]
    kd, ks, pe, pi = T("kernel_dispatch"), T("kernel_symbol"), T("rocpd_pmc_event"), T("rocpd_info_pmc")
    pid = [r[0] for r in cur.execute(f"select id from {pi} where name='{counter}'")]
    names = {r[0]: r[1] for r in cur.execute(f"select id, kernel_name from {ks}")}
    vals = collections.defaultdict(float)
    for ev, v in cur.execute(f"select event_id, value from {pe} where pmc_id in ({','.join(map(str, pid))})"):
        vals[ev] += v
    out = collections.defaultdict(lambda: [0, 0.0])
    for kid, ev in cur.execute(f"select kernel_id, event_id from {kd}"):
        fam = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", names[kid])
        fam = re.sub(r"(_kernel|I[a-zL].*|E[vP].*)$", "", re.split(r"I(?=[Lt])", fam)[0])
        o = out[fam]; o[0] += 1; o[1] += vals.get(ev, 0.0)
    return out


if __name__ == "__main__":
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    res = {}
    for k in sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch.get(k, [0, 0])[1] + write.get(k, [0, 0])[1])):
        n = max(fetch.get(k, [0, 0])[0], write.get(k, [0, 0])[0])
        rd = 2.0 * fetch.get(k, [0, 0])[1] * 1024 / max(n, 1)
        wr = write.get(k, [0, 0])[1] * 1024 / max(n, 1)
        res[k] = {"launches": n, "read_MB_per_launch": rd / 1e6, "write_MB_per_launch": wr / 1e6}
        print("%-28s launches %5d  read %9.2f MB  write %9.2f MB  per launch" % (k[:28], n, rd / 1e6, wr / 1e6))
    if len(sys.argv) > 3:
        json.dump(res, open(sys.argv[3], "w"), indent=1)
