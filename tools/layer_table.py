#!/usr/bin/env python3
"""Per-layer table of a recorded training plan: traces the model on the CPU (record mode never touches the device) and,
when a plan profile written by `bench.py --profile-out` is given, joins every convolution launch with its measured time.

    python tools/layer_table.py [--model yolox_s] [--size 640] [--batch 32] [--profile profiles/rNN_plan_profile.json]
"""
import argparse
import json
import os
import sys

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pl_yolo_amd  # noqa: E402
from pl_yolo_amd import graph as G  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="yolox_s")
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--profile", default=None)
    a = ap.parse_args()
    fam = "yolov7" if a.model.startswith("yolov7") else "yolox"
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "model", fam, a.model + ".yaml")))
    torch.manual_seed(96)
    model = pl_yolo_amd.build_model(cfg, 80).train()
    model.compute_dtype = "bf16"
    r = model.runner()
    dev = torch.device("cpu")
    r.adopt(dev)
    s = r._build(a.batch, a.size, a.size, 100, "train", dev)
    convs = []
    for op in s.g.ops:
        if isinstance(op, (G.ConvUnitOp, G.ConvPairOp)):
            d = op.desc
            convs.append((type(op).__name__, d.ksize, d.stride, d.Cin, d.Cout, d.H, d.W))
        elif isinstance(op, G.HeadPredOp):
            for d in (op.d_ro, op.d_cls):
                convs.append(("HeadPred", d.ksize, d.stride, d.Cin, d.Cout, d.H, d.W))
    times = None
    if a.profile:
        prof = json.load(open(a.profile))
        fwd = [o for o in prof["ops"] if o[1].startswith("conv_mfma_fwd") or o[1].startswith("conv_pw_fwd")]
        times = [o[2] * 1e3 for o in fwd]
    tot = {}
    for i, c in enumerate(convs):
        name, k, st, ci, co, H, W = c
        M = a.batch * (H // st) * (W // st)
        gf = 2.0 * M * ci * co * k * k / 1e9
        mb = (a.batch * H * W * ci + M * co) * 2 / 1e6
        ideal = max(gf / 2.5e6, mb / 8e6) * 1e3
        t = times[i] if times and i < len(times) else float("nan")
        key = "k%d s%d" % (k, st)
        tt = tot.setdefault(key, [0, 0.0, 0.0])
        tt[0] += 1; tt[1] += t; tt[2] += ideal
        print("%3d %-10s k%d s%d %4d->%4d @%3dx%-3d  %7.2f GF %7.1f MB  ideal %6.1f us  fwd %6.1f us" % (i, name, k, st, ci, co, H, W, gf, mb, ideal, t))
    for k, v in tot.items():
        print(k, "n=%d fwd total %.1f us, roofline %.1f us" % (v[0], v[1], v[2]))


if __name__ == "__main__":
    main()
