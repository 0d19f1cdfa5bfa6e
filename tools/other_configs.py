#!/usr/bin/env python3
"""The other BASELINE configs on one MI355X (GPU box): bench line + per-family launch times -> a markdown table.
   python tools/other_configs.py out.md [ref]      (ref: directory of the other checkout of the same-box A/B, default the newest of _r2, _r1)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1]
REF = sys.argv[2] if len(sys.argv) > 2 else next((d for d in ("_r2", "_r1") if os.path.isdir(os.path.join(ROOT, d))), "_r2")
CONFIGS = [("YOLOv7 640² B=32", ["--model", "yolov7", "--batch", "32"]),
           ("YOLOX-l 640² B=16", ["--model", "yolox_l", "--batch", "16"]),
           ("YOLOX-x 1280² B=16", ["--model", "yolox_x", "--size", "1280", "--batch", "16"])]


def run(tree, args, prof=None):
    cmd = [sys.executable, "bench.py", "--no-cpu-baseline", "--steps", "20", "--warmup", "3"] + args + (["--profile-out", prof] if prof else [])
    p = subprocess.run(cmd, cwd=tree, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return json.loads(lines[-1]) if lines else None


md = ["# Other BASELINE configs (one MI355X, bf16, synthetic): same-box A/B against the tree in %s/, and the per-family" % REF,
      "# serialised launch time of one step of the current build", ""]
for name, args in CONFIGS:
    prof = "/tmp/prof.json"
    rows = []
    for rep in range(2):
        a = run(os.path.join(ROOT, REF), args) if os.path.isdir(os.path.join(ROOT, REF)) else None
        b = run(ROOT, args, prof)
        rows.append((a, b))
    fmt = lambda d: "n/a" if d is None else "%.0f img/s, %.2f ms" % (d["value"], d["ms_per_step"])
    md.append("## %s — %s tree: %s — current: %s" % (name, REF, " / ".join(fmt(a) for a, _ in rows), " / ".join(fmt(b) for _, b in rows)))
    md.append("")
    try:
        pr = json.load(open(prof))
        md.append("| family | launches | ms | TFLOP/s | GB/s (algorithmic) |")
        md.append("|---|---|---|---|---|")
        for r in sorted(pr["rows"], key=lambda r: -r[2])[:12]:
            ms = r[2]
            md.append("| `%s` | %d | %.3f | %s | %s |" % (r[0], r[1], ms, ("%.0f" % (r[3] / ms / 1e9)) if r[3] else "–", ("%.0f" % (r[4] / ms / 1e6)) if r[4] else "–"))
        md.append("")
        md.append("sum of stand-alone launch times %.2f ms" % pr["sum_ms"])
        md.append("")
    except Exception as e:   # noqa
        md.append("(no profile: %r)" % (e,))
open(out, "w").write("\n".join(md) + "\n")
print("\n".join(md))
