#!/usr/bin/env python3
"""Where does a lane's wall time go?  (GPU box)  One eager multi-lane replay of the backward (or forward) plan with a timing event
behind every K-th launch of the lane: per segment, the wall time against the sum of the stand-alone times of its launches (plan
profile of the same session) -- the difference is what the lane waits: kernel boundaries, cross-lane events, the other lanes' traffic.
   python tools/lane_stamps.py [bwd|fwd] [lane] [K]"""
import ctypes as C, os, sys
import torch, yaml
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, pl_yolo_amd
from pl_yolo_amd import _lib
which = sys.argv[1] if len(sys.argv) > 1 else "bwd"
lane = int(sys.argv[2]) if len(sys.argv) > 2 else 0
K = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device("cuda:0")
cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "model", "yolox", "yolox_s.yaml")))
torch.manual_seed(96)
model = pl_yolo_amd.build_model(cfg, 80); model.compute_dtype = "bf16"; model = model.to(dev).train()
imgs, labels = bench.synthetic(32, 640, 80, 1234)
imgs, labels = imgs.to(dev), labels.to(dev)
for _ in range(5):
    out = model(imgs, labels); model.zero_grad(set_to_none=True); out["loss"].backward()
torch.cuda.synchronize()
r = model.runner()
s = [v for k, v in r.sessions.items() if k[4] == "train"][0]
st = torch.cuda.current_stream().cuda_stream
plan = s.bwd if which == "bwd" else s.fwd
r._focus(s, imgs)
if which == "bwd":
    s.fwd.run(st)
prof = plan.profile(st)      # stand-alone time of every launch, recorded order (markers excluded)
# map recorded op index -> profile row: profile() skips record / wait markers
n = plan.size()
buf = C.create_string_buffer(96); fl, by = C.c_double(), C.c_double()
rows, k = {}, 0
for i in range(n):
    _lib.call("plyolo_plan_op_info", plan.h, i, buf, 96, C.byref(fl), C.byref(by))
    if buf.value not in (b"record", b"wait"):
        rows[i] = prof[k]; k += 1
best = None
for rep in range(4):
    r._focus(s, imgs)
    if which == "bwd":
        s.fwd.run(st)
    torch.cuda.synchronize()
    ms = (C.c_float * 512)(); at = (C.c_int * 512)()
    cnt = _lib.lib().plyolo_plan_stamp_times(plan.h, st, lane, K, C.cast(ms, C.c_void_p), C.cast(at, C.c_void_p), 512)
    if rep:
        best = ([float(ms[i]) for i in range(cnt)], [int(at[i]) for i in range(cnt)])
tms, ats = best
prev_t, prev_i = 0.0, -1
tot_wall = tot_alone = 0.0
print("%s plan, lane %d, a stamp every %d launches" % (which, lane, K))
for t, i in zip(tms, ats):
    seg = [rows[j] for j in range(prev_i + 1, i + 1) if j in rows and rows[j][4] == lane]
    alone = sum(x[1] for x in seg)
    names = {}
    for x in seg:
        names[x[0].split("<")[0]] = names.get(x[0].split("<")[0], 0) + 1
    mb = sum(x[3] for x in seg) / 1e6
    print("  %6.3f -> %6.3f ms  wall %5.0f us  stand-alone %5.0f us (%4.0f MB)  %s" % (prev_t, t, (t - prev_t) * 1e3, alone * 1e3, mb,
          " ".join("%s x%d" % (k_, v) for k_, v in names.items())))
    tot_wall += t - prev_t; tot_alone += alone
    prev_t, prev_i = t, i
print("  total wall %.3f ms, stand-alone %.3f ms (each stand-alone time includes the profile mode's ~4 us event pair)" % (tot_wall, tot_alone))
