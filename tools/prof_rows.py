"""Per-family summary of a plan profile written by bench.py --profile-out: launches, total ms, algorithmic TB/s / TFLOP/s."""
import json, sys
p = json.load(open(sys.argv[1]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
print("  sum of stand-alone launch times %.3f ms, %d launches" % (p["sum_ms"], len(p["ops"])))
for k, l, ms, fl, by in p["rows"][:n]:
    print("  %-44s %3d  %7.3f ms  %6.2f TB/s  %7.1f TF/s" % (k, l, ms, by / ms / 1e9 if ms else 0, fl / ms / 1e9 if ms else 0))

# per plan and lane: launches and the sum of their stand-alone times (the profile mode's event pair, ~4 us, is inside every launch)
if p["ops"] and len(p["ops"][0]) > 5:
    import collections
    acc = collections.OrderedDict()
    for plan, k, ms, fl, by, lane in p["ops"]:
        a = acc.setdefault((plan, lane), [0, 0.0, 0.0])
        a[0] += 1; a[1] += ms; a[2] += by
    for (plan, lane), (n_, ms, by) in acc.items():
        print("  %s lane %d: %3d launches, %.3f ms stand-alone, %.2f GB algorithmic" % (plan, lane, n_, ms, by / 1e9))
